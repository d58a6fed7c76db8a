// Chess board, move generation and make-move for the native search driver.
//
// Restates what the search needs from batch_first/numba_board.py (python-chess derived bitboard
// code under Numba): the node record's board fields (classes_and_structs.py:7-29), push_moves
// (numba_board.py:242-417), legal move generation (set_up_move_array :1058-1082, has_legal_move
// :1089-1107, is_legal_move :1149-1151), has_insufficient_material (:197-217) and
// convert_board_to_whites_perspective (:5-31).  Conventions as in batch_first/__init__.py:176-205:
// a1 = bit 0, occupied_co[0] = black, [1] = white, turn 1 = white, piece types P..K = 1..6,
// castling_rights = bitboard of the rooks that may still castle, ep_square 0 = none and set after
// EVERY double pawn push (python-chess semantics, numba_board.py:352-353), castling moves written
// as the king's two-square move (e1g1), promotions as piece type in the third byte.
//
// Not restated: the Zobrist keys.  The reference uses python-chess's POLYGLOT_RANDOM_ARRAY
// (batch_first/__init__.py), which is not available offline; keys here come from splitmix64 with
// the same key layout (piece x colour x square, 4 castling, 8 ep files, turn) and the same
// incremental update rules, so hashes differ in value from the reference's but not in meaning.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace bfs {

typedef uint64_t u64;

enum { PAWN = 1, KNIGHT, BISHOP, ROOK, QUEEN, KING };
enum { BLACK = 0, WHITE = 1 };

struct Move {
    uint8_t from, to, promo;
    bool operator==(const Move& o) const { return from == o.from && to == o.to && promo == o.promo; }
};
constexpr Move kNoMove = {255, 255, 255};
constexpr int kMaxMoves = 100;  // MAX_MOVES_LOOKED_AT, batch_first/__init__.py:126

struct Board {
    u64 pawns, knights, bishops, rooks, queens, kings;
    u64 occ_co[2];
    u64 occ;
    u64 castling;
    u64 hash;
    int8_t turn;
    uint8_t ep;
    uint8_t halfmove;
};

// Bytes 0..81 of numpy_node_info_dtype, what the evaluation library reads (include/batchfirst.h).
#pragma pack(push, 1)
struct EvalRecord {
    u64 pawns, knights, bishops, rooks, queens, kings, occ_black, occ_white, occ;
    int8_t turn;
    u64 castling;
    uint8_t ep;
    uint8_t pad[6];
};
#pragma pack(pop)
static_assert(sizeof(EvalRecord) == 88, "compact record stride");

void init_tables();  // idempotent, thread-safe after the first call

bool board_from_fen(const std::string& fen, Board* out);
std::string board_to_fen(const Board& b);
u64 compute_hash(const Board& b);
void to_record(const Board& b, EvalRecord* r);
void from_record(const EvalRecord& r, Board* b);  // hash recomputed, halfmove 0

int piece_type_at(const Board& b, int sq);
u64 attackers(const Board& b, int color, int sq, u64 occ);
bool in_check(const Board& b);

// Legal moves of the side to move, at most kMaxMoves (further ones are dropped, as the reference's
// fixed-size array does).  Returns the count.
int legal_moves(const Board& b, Move* out);
bool has_legal_move(const Board& b);
bool is_legal(const Board& b, Move m);
bool is_legal_direct(const Board& b, Move m);  // decided from the move alone (no generator); must agree with is_legal
void push(Board& b, Move m);
bool insufficient_material(const Board& b);
void to_whites_perspective(Board& b);

uint64_t perft(const Board& b, int depth);

}  // namespace bfs
