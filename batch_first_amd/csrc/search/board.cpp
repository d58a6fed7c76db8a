// See board.h.  Plain bitboards: ray tables with first-blocker lookup for sliders; legal moves =
// pseudo-legal moves filtered by king safety (moves that cannot expose the king are passed
// without making them).
#include "board.h"

#include <mutex>
#include <sstream>

namespace bfs {

namespace {

u64 g_knight[64], g_king[64], g_pawn_att[2][64], g_ray[8][64], g_lines[64];
u64 g_keys[781];
std::once_flag g_once;

enum { N = 0, E, NE, NW, S, W, SE, SW };  // 0..3 run towards higher squares, 4..7 towards lower
const int kDr[8] = {1, 0, 1, 1, -1, 0, -1, -1};
const int kDf[8] = {0, 1, 1, -1, 0, -1, 1, -1};

inline u64 bit(int sq) { return 1ull << sq; }
inline int lsb(u64 b) { return __builtin_ctzll(b); }
inline int msb(u64 b) { return 63 - __builtin_clzll(b); }
inline int pop_lsb(u64& b) {
    int s = lsb(b);
    b &= b - 1;
    return s;
}

void build_tables() {
    for (int sq = 0; sq < 64; ++sq) {
        int r = sq >> 3, f = sq & 7;
        auto add = [&](u64& m, int rr, int ff) {
            if (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) m |= bit(rr * 8 + ff);
        };
        g_knight[sq] = g_king[sq] = g_pawn_att[0][sq] = g_pawn_att[1][sq] = 0;
        const int kn[8][2] = {{1, 2}, {2, 1}, {-1, 2}, {-2, 1}, {1, -2}, {2, -1}, {-1, -2}, {-2, -1}};
        for (auto& d : kn) add(g_knight[sq], r + d[0], f + d[1]);
        for (int dr = -1; dr <= 1; ++dr)
            for (int df = -1; df <= 1; ++df)
                if (dr || df) add(g_king[sq], r + dr, f + df);
        add(g_pawn_att[WHITE][sq], r + 1, f - 1);
        add(g_pawn_att[WHITE][sq], r + 1, f + 1);
        add(g_pawn_att[BLACK][sq], r - 1, f - 1);
        add(g_pawn_att[BLACK][sq], r - 1, f + 1);
        g_lines[sq] = 0;
        for (int d = 0; d < 8; ++d) {
            u64 m = 0;
            int rr = r + kDr[d], ff = f + kDf[d];
            while (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
                m |= bit(rr * 8 + ff);
                rr += kDr[d];
                ff += kDf[d];
            }
            g_ray[d][sq] = m;
            g_lines[sq] |= m;
        }
    }
    u64 x = 0x9E3779B97F4A7C15ull;  // splitmix64
    for (auto& k : g_keys) {
        x += 0x9E3779B97F4A7C15ull;
        u64 z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        k = z ^ (z >> 31);
    }
}

inline u64 ray_attacks(int d, int sq, u64 occ) {
    u64 r = g_ray[d][sq];
    u64 blk = r & occ;
    if (blk) r ^= g_ray[d][d < 4 ? lsb(blk) : msb(blk)];
    return r;
}
inline u64 rook_attacks(int sq, u64 occ) {
    return ray_attacks(N, sq, occ) | ray_attacks(S, sq, occ) | ray_attacks(E, sq, occ) | ray_attacks(W, sq, occ);
}
inline u64 bishop_attacks(int sq, u64 occ) {
    return ray_attacks(NE, sq, occ) | ray_attacks(NW, sq, occ) | ray_attacks(SE, sq, occ) | ray_attacks(SW, sq, occ);
}

inline u64* piece_bb(Board& b, int pt) {
    switch (pt) {
        case PAWN: return &b.pawns;
        case KNIGHT: return &b.knights;
        case BISHOP: return &b.bishops;
        case ROOK: return &b.rooks;
        case QUEEN: return &b.queens;
        default: return &b.kings;
    }
}
inline u64 piece_key(int pt, int color, int sq) { return g_keys[((pt - 1) * 2 + color) * 64 + sq]; }

// the ep file enters the hash only when a pawn of the side to move stands next to the pushed pawn
// (numba_board.py:274-281, :407-413)
inline u64 ep_key(const Board& b) {
    if (!b.ep) return 0;
    u64 m = b.turn ? bit(b.ep) >> 8 : bit(b.ep) << 8;
    u64 adj = ((m << 1) & ~0x0101010101010101ull) | ((m >> 1) & ~0x8080808080808080ull);
    return (adj & b.pawns & b.occ_co[b.turn]) ? g_keys[772 + (b.ep & 7)] : 0;
}
inline u64 castling_keys(u64 rights) {
    u64 h = 0;
    if (rights & bit(7)) h ^= g_keys[768];
    if (rights & bit(0)) h ^= g_keys[769];
    if (rights & bit(63)) h ^= g_keys[770];
    if (rights & bit(56)) h ^= g_keys[771];
    return h;
}

void remove_piece(Board& b, int sq, int pt, int color) {
    u64 m = bit(sq);
    *piece_bb(b, pt) ^= m;
    b.occ ^= m;
    b.occ_co[color] ^= m;
    b.hash ^= piece_key(pt, color, sq);
}
void set_piece(Board& b, int sq, int pt, int color) {
    u64 m = bit(sq);
    *piece_bb(b, pt) |= m;
    b.occ |= m;
    b.occ_co[color] |= m;
    b.hash ^= piece_key(pt, color, sq);
}

void add_move(Move* out, int& n, int from, int to, int promo = 0) {
    if (n < kMaxMoves) out[n] = Move{(uint8_t)from, (uint8_t)to, (uint8_t)promo};
    ++n;
}

// Pseudo-legal moves (castling fully checked).  Returns the count generated (may exceed kMaxMoves;
// only the first kMaxMoves are stored).
int pseudo_legal(const Board& b, Move* out) {
    int n = 0;
    const int us = b.turn, them = us ^ 1;
    const u64 own = b.occ_co[us], opp = b.occ_co[them], occ = b.occ;
    // pieces in ascending square order, targets ascending
    u64 pieces = own & ~b.pawns;
    while (pieces) {
        int from = pop_lsb(pieces);
        u64 m = bit(from), att;
        if (b.knights & m) att = g_knight[from];
        else if (b.kings & m) att = g_king[from];
        else {
            att = 0;
            if ((b.bishops | b.queens) & m) att |= bishop_attacks(from, occ);
            if ((b.rooks | b.queens) & m) att |= rook_attacks(from, occ);
        }
        att &= ~own;
        while (att) add_move(out, n, from, pop_lsb(att));
    }
    // castling (standard chess: king on e1/e8, rights bit on the rook's corner)
    const int ksq0 = us ? 4 : 60;
    if ((b.kings & own & bit(ksq0)) && (b.castling & (us ? 0xFFull : 0xFFull << 56))) {
        auto safe = [&](int s) { return attackers(b, them, s, occ) == 0; };
        const u64 rook_k = bit(ksq0 + 3), rook_q = bit(ksq0 - 4);
        if ((b.castling & rook_k) && (b.rooks & own & rook_k) && !(occ & (bit(ksq0 + 1) | bit(ksq0 + 2))) &&
            safe(ksq0) && safe(ksq0 + 1) && safe(ksq0 + 2))
            add_move(out, n, ksq0, ksq0 + 2);
        if ((b.castling & rook_q) && (b.rooks & own & rook_q) &&
            !(occ & (bit(ksq0 - 1) | bit(ksq0 - 2) | bit(ksq0 - 3))) && safe(ksq0) && safe(ksq0 - 1) && safe(ksq0 - 2))
            add_move(out, n, ksq0, ksq0 - 2);
    }
    // pawns
    u64 pawns = b.pawns & own;
    const int fwd = us ? 8 : -8;
    const int promo_rank = us ? 7 : 0, start_rank = us ? 1 : 6;
    while (pawns) {
        int from = pop_lsb(pawns);
        u64 caps = g_pawn_att[us][from] & opp;
        while (caps) {
            int to = pop_lsb(caps);
            if ((to >> 3) == promo_rank)
                for (int p = QUEEN; p >= KNIGHT; --p) add_move(out, n, from, to, p);
            else add_move(out, n, from, to);
        }
        int to = from + fwd;
        if (to >= 0 && to < 64 && !(occ & bit(to))) {
            if ((to >> 3) == promo_rank)
                for (int p = QUEEN; p >= KNIGHT; --p) add_move(out, n, from, to, p);
            else {
                add_move(out, n, from, to);
                if ((from >> 3) == start_rank && !(occ & bit(to + fwd))) add_move(out, n, from, to + fwd);
            }
        }
        if (b.ep && (g_pawn_att[us][from] & bit(b.ep)) && !(occ & bit(b.ep)) && (b.ep >> 3) == (us ? 5 : 2) &&
            (b.pawns & opp & bit(b.ep - fwd)))
            add_move(out, n, from, b.ep);
    }
    return n;
}

bool leaves_king_safe(const Board& b, Move m, int ksq, bool checked) {
    const int us = b.turn;
    const bool is_king = m.from == ksq;
    const bool is_ep = (b.pawns & bit(m.from)) && m.to == b.ep && b.ep && !(b.occ & bit(m.to));
    if (!checked && !is_king && !is_ep && !(g_lines[ksq] & bit(m.from))) return true;
    if (is_king && ((int)m.to - (int)m.from == 2 || (int)m.from - (int)m.to == 2))
        return true;  // castling: every square of the king's path was checked when the move was generated
    Board c = b;
    push(c, m);
    int k = lsb(c.kings & c.occ_co[us]);
    return attackers(c, us ^ 1, k, c.occ) == 0;
}

}  // namespace

void init_tables() { std::call_once(g_once, build_tables); }

int piece_type_at(const Board& b, int sq) {
    u64 m = bit(sq);
    if (!(b.occ & m)) return 0;
    if (b.pawns & m) return PAWN;
    if (b.knights & m) return KNIGHT;
    if (b.bishops & m) return BISHOP;
    if (b.rooks & m) return ROOK;
    if (b.queens & m) return QUEEN;
    return KING;
}

u64 attackers(const Board& b, int color, int sq, u64 occ) {
    u64 a = (g_knight[sq] & b.knights) | (g_king[sq] & b.kings) | (g_pawn_att[color ^ 1][sq] & b.pawns);
    a |= rook_attacks(sq, occ) & (b.rooks | b.queens);
    a |= bishop_attacks(sq, occ) & (b.bishops | b.queens);
    return a & b.occ_co[color] & occ;
}

bool in_check(const Board& b) {
    u64 k = b.kings & b.occ_co[b.turn];
    return k && attackers(b, b.turn ^ 1, lsb(k), b.occ) != 0;
}

u64 compute_hash(const Board& b) {
    u64 h = 0;
    for (int sq = 0; sq < 64; ++sq) {
        int pt = piece_type_at(b, sq);
        if (pt) h ^= piece_key(pt, (b.occ_co[WHITE] >> sq) & 1, sq);
    }
    h ^= castling_keys(b.castling);
    h ^= ep_key(b);
    if (b.turn) h ^= g_keys[780];
    return h;
}

int legal_moves(const Board& b, Move* out) {
    Move tmp[256];
    // pseudo_legal stores at most kMaxMoves; generate into a big buffer through a local copy of the cap
    int n = pseudo_legal(b, tmp);
    if (n > kMaxMoves) n = kMaxMoves;
    u64 kbb = b.kings & b.occ_co[b.turn];
    if (!kbb) {
        std::memcpy(out, tmp, n * sizeof(Move));
        return n;
    }
    const int ksq = lsb(kbb);
    const bool checked = attackers(b, b.turn ^ 1, ksq, b.occ) != 0;
    int k = 0;
    for (int i = 0; i < n; ++i)
        if (leaves_king_safe(b, tmp[i], ksq, checked)) out[k++] = tmp[i];
    return k;
}

bool has_legal_move(const Board& b) {
    Move tmp[256];
    int n = pseudo_legal(b, tmp);
    if (n > kMaxMoves) n = kMaxMoves;
    u64 kbb = b.kings & b.occ_co[b.turn];
    if (!kbb) return n > 0;
    const int ksq = lsb(kbb);
    const bool checked = attackers(b, b.turn ^ 1, ksq, b.occ) != 0;
    for (int i = 0; i < n; ++i)
        if (leaves_king_safe(b, tmp[i], ksq, checked)) return true;
    return false;
}

bool is_legal(const Board& b, Move m) {
    if (m.from > 63 || m.to > 63) return false;
    Move tmp[kMaxMoves];
    int n = legal_moves(b, tmp);
    for (int i = 0; i < n; ++i)
        if (tmp[i] == m) return true;
    return false;
}

// Legality of one move decided from the move itself, without the generator: geometry of the piece
// on the from-square by walking squares (no attack tables), castling and en-passant conditions
// spelled out, then the move is made on a copy and the mover's king must not be attacked.  The
// counterpart of the reference's is_legal_move = is_pseudo_legal_move and not is_into_check
// (numba_board.py:1109-1151); the sweep of code_testing.py:266-293 holds it against legal_moves().
bool is_legal_direct(const Board& b, Move m) {
    if (m.from > 63 || m.to > 63 || m.from == m.to) return false;
    const int us = b.turn, them = us ^ 1;
    const u64 own = b.occ_co[us], opp = b.occ_co[them];
    if (!(own & bit(m.from)) || (own & bit(m.to))) return false;
    const int pt = piece_type_at(b, m.from);
    const int fr = m.from >> 3, ff = m.from & 7, tr = m.to >> 3, tf = m.to & 7;
    const int dr = tr - fr, df = tf - ff, adr = dr < 0 ? -dr : dr, adf = df < 0 ? -df : df;
    auto path_clear = [&]() {   // squares strictly between from and to on a rank, file or diagonal
        const int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
        for (int r = fr + sr, f = ff + sf; r != tr || f != tf; r += sr, f += sf)
            if (b.occ & bit(r * 8 + f)) return false;
        return true;
    };
    if (m.promo && (pt != PAWN || m.promo < KNIGHT || m.promo > QUEEN)) return false;
    bool castle = false;
    switch (pt) {
        case PAWN: {
            const int step = us ? 1 : -1, last = us ? 7 : 0, home = us ? 1 : 6;
            if ((tr == last) != (m.promo != 0)) return false;
            if (df == 0) {
                if (b.occ & bit(m.to)) return false;
                if (dr == step) break;
                if (dr == 2 * step && fr == home && !(b.occ & bit((fr + step) * 8 + ff))) break;
                return false;
            }
            if (adf != 1 || dr != step) return false;
            if (opp & bit(m.to)) break;
            // en passant: onto the empty ep square, with the pawn that just made its double step behind it
            if (b.ep && m.to == b.ep && tr == (us ? 5 : 2) && !(b.occ & bit(m.to)) &&
                (b.pawns & opp & bit((tr - step) * 8 + tf)))
                break;
            return false;
        }
        case KNIGHT:
            if (!((adr == 1 && adf == 2) || (adr == 2 && adf == 1))) return false;
            break;
        case BISHOP:
            if (adr != adf || !path_clear()) return false;
            break;
        case ROOK:
            if ((dr != 0 && df != 0) || !path_clear()) return false;
            break;
        case QUEEN:
            if (!((adr == adf) || dr == 0 || df == 0) || !path_clear()) return false;
            break;
        default:  // king
            if (adr <= 1 && adf <= 1) break;
            if (dr != 0 || adf != 2 || m.from != (us ? 4 : 60)) return false;
            castle = true;
            break;
    }
    if (castle) {
        const int rook_sq = df > 0 ? m.from + 3 : m.from - 4;
        if (!(b.castling & bit(rook_sq)) || !(b.rooks & own & bit(rook_sq))) return false;
        const int lo = rook_sq < m.from ? rook_sq : m.from, hi = rook_sq < m.from ? m.from : rook_sq;
        for (int s = lo + 1; s < hi; ++s)
            if (b.occ & bit(s)) return false;
        const int stepf = df > 0 ? 1 : -1;
        for (int s = m.from; s != m.to + stepf; s += stepf)   // start, crossed and target square
            if (attackers(b, them, s, b.occ)) return false;
        return true;
    }
    Board c = b;
    push(c, m);
    const u64 k = c.kings & c.occ_co[us];
    return !k || attackers(c, them, lsb(k), c.occ) == 0;
}

void push(Board& b, Move m) {
    const int us = b.turn, them = us ^ 1;
    b.hash ^= ep_key(b);
    const int old_ep = b.ep;
    b.ep = 0;
    b.hash ^= castling_keys(b.castling);
    const u64 from_bb = bit(m.from), to_bb = bit(m.to);
    int pt = piece_type_at(b, m.from);
    const int captured = (b.occ_co[them] & to_bb) ? piece_type_at(b, m.to) : 0;
    b.halfmove = (pt == PAWN || captured) ? 0 : (uint8_t)(b.halfmove + 1);
    b.castling &= ~to_bb & ~from_bb;
    if (pt == KING) b.castling &= us ? ~0xFFull : ~(0xFFull << 56);
    remove_piece(b, m.from, pt, us);
    if (captured) remove_piece(b, m.to, captured, them);
    if (pt == PAWN) {
        int diff = (int)m.to - (int)m.from;
        if (diff == 16 || diff == -16) b.ep = (uint8_t)(m.from + diff / 2);
        else if (old_ep && m.to == old_ep && !captured && (diff == 7 || diff == 9 || diff == -7 || diff == -9))
            remove_piece(b, us ? old_ep - 8 : old_ep + 8, PAWN, them);
    }
    if (m.promo) pt = m.promo;
    if (pt == KING && ((int)m.to - (int)m.from == 2 || (int)m.from - (int)m.to == 2) && (m.from == 4 || m.from == 60)) {
        const bool kingside = m.to > m.from;
        const int rook_from = kingside ? m.from + 3 : m.from - 4, rook_to = kingside ? m.from + 1 : m.from - 1;
        if (b.rooks & b.occ_co[us] & bit(rook_from)) {
            remove_piece(b, rook_from, ROOK, us);
            set_piece(b, rook_to, ROOK, us);
        }
    }
    set_piece(b, m.to, pt, us);
    b.turn = (int8_t)them;
    b.hash ^= g_keys[780];
    b.hash ^= castling_keys(b.castling);
    b.hash ^= ep_key(b);
}

bool insufficient_material(const Board& b) {
    if (b.pawns || b.rooks || b.queens) return false;
    if (__builtin_popcountll(b.occ) <= 3) return true;
    if (b.knights) return false;
    const u64 dark = 0xAA55AA55AA55AA55ull, light = ~dark;
    if (!(b.bishops & dark)) return true;
    if (!(b.bishops & light)) return true;
    return false;
}

void to_whites_perspective(Board& b) {
    if (b.turn) return;
    auto flip = [](u64 x) { return __builtin_bswap64(x); };
    u64 w = flip(b.occ_co[BLACK]), k = flip(b.occ_co[WHITE]);
    b.occ_co[WHITE] = w;
    b.occ_co[BLACK] = k;
    b.occ = flip(b.occ);
    b.pawns = flip(b.pawns); b.knights = flip(b.knights); b.bishops = flip(b.bishops);
    b.rooks = flip(b.rooks); b.queens = flip(b.queens); b.kings = flip(b.kings);
    b.castling = flip(b.castling);
    if (b.ep) b.ep ^= 56;
    b.turn = 1;
    b.hash = compute_hash(b);
}

void to_record(const Board& b, EvalRecord* r) {
    r->pawns = b.pawns; r->knights = b.knights; r->bishops = b.bishops; r->rooks = b.rooks;
    r->queens = b.queens; r->kings = b.kings; r->occ_black = b.occ_co[0]; r->occ_white = b.occ_co[1];
    r->occ = b.occ; r->turn = b.turn; r->castling = b.castling; r->ep = b.ep;
    std::memset(r->pad, 0, sizeof r->pad);
}

void from_record(const EvalRecord& r, Board* b) {
    b->pawns = r.pawns; b->knights = r.knights; b->bishops = r.bishops; b->rooks = r.rooks;
    b->queens = r.queens; b->kings = r.kings; b->occ_co[0] = r.occ_black; b->occ_co[1] = r.occ_white;
    b->occ = r.occ; b->turn = r.turn; b->castling = r.castling; b->ep = r.ep; b->halfmove = 0;
    b->hash = compute_hash(*b);
}

bool board_from_fen(const std::string& fen, Board* out) {
    init_tables();
    Board b;
    std::memset(&b, 0, sizeof b);
    std::istringstream ss(fen);
    std::string placement, side, castle, ep;
    int half = 0, full = 1;
    if (!(ss >> placement >> side)) return false;
    if (!(ss >> castle)) castle = "-";
    if (!(ss >> ep)) ep = "-";
    if (!(ss >> half)) half = 0;
    if (!(ss >> full)) full = 1;
    int r = 7, f = 0;
    for (char c : placement) {
        if (c == '/') { --r; f = 0; continue; }
        if (c >= '1' && c <= '8') { f += c - '0'; continue; }
        if (r < 0 || f > 7) return false;
        int color = (c >= 'A' && c <= 'Z') ? WHITE : BLACK;
        char l = (char)(c | 0x20);
        int pt = l == 'p' ? PAWN : l == 'n' ? KNIGHT : l == 'b' ? BISHOP : l == 'r' ? ROOK : l == 'q' ? QUEEN : l == 'k' ? KING : 0;
        if (!pt) return false;
        u64 m = bit(r * 8 + f);
        *piece_bb(b, pt) |= m;
        b.occ |= m;
        b.occ_co[color] |= m;
        ++f;
    }
    b.turn = side == "w" ? 1 : 0;
    for (char c : castle) {
        if (c == 'K') b.castling |= bit(7);
        else if (c == 'Q') b.castling |= bit(0);
        else if (c == 'k') b.castling |= bit(63);
        else if (c == 'q') b.castling |= bit(56);
    }
    b.castling &= b.rooks;
    if (ep.size() == 2 && ep[0] >= 'a' && ep[0] <= 'h' && ep[1] >= '1' && ep[1] <= '8') b.ep = (uint8_t)((ep[1] - '1') * 8 + (ep[0] - 'a'));
    b.halfmove = (uint8_t)(half > 255 ? 255 : half);
    b.hash = compute_hash(b);
    *out = b;
    return true;
}

std::string board_to_fen(const Board& b) {
    std::string s;
    for (int r = 7; r >= 0; --r) {
        int empty = 0;
        for (int f = 0; f < 8; ++f) {
            int sq = r * 8 + f, pt = piece_type_at(b, sq);
            if (!pt) { ++empty; continue; }
            if (empty) { s += (char)('0' + empty); empty = 0; }
            char c = " pnbrqk"[pt];
            if (b.occ_co[WHITE] & bit(sq)) c = (char)(c - 32);
            s += c;
        }
        if (empty) s += (char)('0' + empty);
        if (r) s += '/';
    }
    s += b.turn ? " w " : " b ";
    std::string c;
    if (b.castling & bit(7)) c += 'K';
    if (b.castling & bit(0)) c += 'Q';
    if (b.castling & bit(63)) c += 'k';
    if (b.castling & bit(56)) c += 'q';
    s += c.empty() ? "-" : c;
    s += ' ';
    if (b.ep) { s += (char)('a' + (b.ep & 7)); s += (char)('1' + (b.ep >> 3)); } else s += '-';
    s += ' ' + std::to_string((int)b.halfmove) + " 1";
    return s;
}

uint64_t perft(const Board& b, int depth) {
    Move mv[kMaxMoves];
    int n = legal_moves(b, mv);
    if (depth <= 1) return depth == 1 ? (uint64_t)n : 1;
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) {
        Board c = b;
        push(c, mv[i]);
        total += perft(c, depth - 1);
    }
    return total;
}

}  // namespace bfs
