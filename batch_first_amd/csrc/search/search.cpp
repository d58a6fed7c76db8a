// See search.h.
#include "search.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include <sys/mman.h>

#include "board.h"
#include "threads.h"

using namespace bfs;

namespace {

constexpr float FMAX = FLT_MAX, FMIN = -FLT_MAX;
constexpr float TIE = 0.0f;                      // TIE_RESULT_SCORE
constexpr int kMaxDepth = 100;                   // MAX_SEARCH_DEPTH
constexpr uint8_t kNoMoreMoves = 255;            // NO_MORE_MOVES_VALUE
constexpr uint8_t kNoEntry = 255;                // NO_TT_ENTRY_VALUE

float g_win[kMaxDepth], g_loss[kMaxDepth], g_tt_move_score;
uint8_t g_filter[64][64][6];
std::once_flag g_const_once;

// batch_first/__init__.py:119-165 (score sentinels) and :51-95 (move filter table)
void build_constants() {
    init_tables();
    const float almost_max = std::nextafterf(FMAX, FMIN), almost_min = std::nextafterf(FMIN, FMAX);
    g_tt_move_score = almost_max;
    g_win[0] = std::nextafterf(almost_max, FMIN);
    g_loss[0] = std::nextafterf(almost_min, FMAX);
    for (int j = 1; j < kMaxDepth; ++j) {
        g_win[j] = std::nextafterf(g_win[j - 1], FMIN);
        g_loss[j] = std::nextafterf(g_loss[j - 1], FMAX);
    }
    std::memset(g_filter, 0, sizeof g_filter);
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            const int fd = (b & 7) - (a & 7), rd = (b >> 3) - (a >> 3);
            int plane = -1;
            if (fd == 0 && rd > 0) plane = rd - 1;
            else if (fd == 0 && rd < 0) plane = -rd + 6;
            else if (rd == 0 && fd > 0) plane = fd + 13;
            else if (rd == 0 && fd < 0) plane = -fd + 20;
            else if (fd == rd && fd > 0) plane = fd + 27;
            else if (fd == -rd && fd > 0) plane = fd + 34;
            else if (fd == -rd && fd < 0) plane = -fd + 41;
            else if (fd == rd && fd < 0) plane = -fd + 48;
            else if (fd == 1 && rd == 2) plane = 56;
            else if (fd == 1 && rd == -2) plane = 57;
            else if (fd == -1 && rd == 2) plane = 58;
            else if (fd == -1 && rd == -2) plane = 59;
            else if (fd == 2 && rd == 1) plane = 60;
            else if (fd == 2 && rd == -1) plane = 61;
            else if (fd == -2 && rd == 1) plane = 62;
            else if (fd == -2 && rd == -1) plane = 63;
            if (plane < 0) continue;
            for (int p = 0; p < 6; ++p) g_filter[a][b][p] = (uint8_t)plane;
            if (rd == 1 && fd >= -1 && fd <= 1)
                for (int p = 2; p < 5; ++p) g_filter[a][b][p] = (uint8_t)(3 * (1 + fd) + 64 + (p - 2));
        }
}

thread_local char g_err[512] = "";
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

struct TTEntry {
    u64 hash = 0;
    float upper = FMAX, lower = FMIN;
    uint8_t depth = kNoEntry;
    Move move = kNoMove;
};

struct Node {
    Node() {}   // deliberately empty: growing the arena must not zero 800 bytes a node; iterate() sets every field it reads
    Board b;
    int parent;
    float separator, best_value;
    uint8_t depth;
    bool terminated;
    uint8_t n_moves, next_move_index, children_left;
    Move prev_move;
    Move moves[kMaxMoves];
    float move_scores[kMaxMoves];
};

// What the tree update reads and writes of an ARENA node, 32 bytes instead of lines scattered over 0.8 KB: the walk
// from a finished child to the root touches these fields only, and in the node itself they share cache lines with
// the move list that the expanding threads write (next_move_index, move_scores), so every hop was a line fetched
// from another core.  The arena node's own best_value / terminated / children_left are not maintained after it
// enters the arena; children in the batch's `fresh` block use their Node fields as before.
struct Hot {
    u64 hash;
    int parent;
    float separator, best_value;
    uint8_t depth, terminated, children_left;
    Move prev_move;
};
static_assert(sizeof(Hot) == 32, "two tree-state entries a cache line");

// The node arena: fixed blocks, so that growing never copies or zeroes what is there (a std::vector of 800-byte
// nodes spent a sixth of the search re-allocating), and an index stays valid for the whole search.
class Arena {
public:
    static constexpr size_t kShift = 14, kBlock = (size_t)1 << kShift;   // 16 384 nodes = 13 MiB a block
    Arena() = default;
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;
    ~Arena() {
        for (Node* b : blocks_) std::free(b);
    }
    size_t size() const { return size_; }
    Node& operator[](size_t i) { return blocks_[i >> kShift][i & (kBlock - 1)]; }
    const Node& operator[](size_t i) const { return blocks_[i >> kShift][i & (kBlock - 1)]; }
    void clear() { size_ = 0; }        // blocks are kept for the next search
    void resize(size_t n) {            // may throw std::bad_alloc; new nodes are uninitialised
        while (blocks_.size() * kBlock < n) {
            if (blocks_.size() == blocks_.capacity()) throw std::bad_alloc();   // reserve_blocks() sized the table
            blocks_.push_back(new_block());
        }
        size_ = n;
    }
    // The table of block pointers never moves after this: operator[] may run on another thread (the tree update)
    // while resize() appends blocks.
    void reserve_blocks(size_t max_nodes) { blocks_.reserve(max_nodes / kBlock + 2); }

private:
    // 2 MiB-aligned and advised as huge pages: a block is touched for the first time while the search runs, and
    // with 4 KiB pages that is 3 300 page faults a block
    static Node* new_block() {
        constexpr size_t kHuge = (size_t)2 << 20;
        const size_t bytes = (kBlock * sizeof(Node) + kHuge - 1) / kHuge * kHuge;
        void* p = std::aligned_alloc(kHuge, bytes);
        if (!p) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
        (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
        return static_cast<Node*>(p);
    }
    std::vector<Node*> blocks_;
    size_t size_ = 0;
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct bfs_search {
    bfs_config cfg;
    std::vector<float> bins_desc;
    std::vector<TTEntry> tt;
    u64 tt_mask = 0;
    Arena nodes;
    // (Insertion into the bins stays on one thread: handing every piece of the pool its share of the bins was
    // measured in round 3 and took twice as long as one thread's appends, 8 against 4 % of a depth-5 search.)
    struct Bin {
        std::vector<int> v;
    };
    std::vector<Bin> bins;
    size_t in_bins = 0;
    // Ancestry of the arena's nodes in four bytes each: (parent + 1) | terminated << 31.  The open list's validity
    // check walks from every candidate to the root; in the 0.8 KB nodes every hop is a cache miss (14 % of a depth-5
    // search), here the upper levels of the tree stay in cache.  Written where a node enters the arena and where
    // the tree update terminates one; read only while no update runs.
    // (calloc: max_nodes entries whose pages are only touched as the arena grows; never reallocated, because the tree
    // update writes entries on the evaluation lane while the caller appends new ones)
    struct FreeDeleter {
        void operator()(uint32_t* p) const { std::free(p); }
    };
    std::unique_ptr<uint32_t[], FreeDeleter> up;
    struct FreeHot {
        void operator()(Hot* p) const { std::free(p); }
    };
    std::unique_ptr<Hot[], FreeHot> hot;   // tree state of the arena's nodes (struct Hot); allocated like `up`
    void enter_tree(int id, const Node& n) {   // a node takes its place in the arena
        Hot& h = hot[id];
        h.hash = n.b.hash;
        h.parent = n.parent;
        h.separator = n.separator;
        h.best_value = n.best_value;
        h.depth = n.depth;
        h.terminated = n.terminated;
        h.children_left = n.children_left;
        h.prev_move = n.prev_move;
        up[id] = (uint32_t)(n.parent + 1) | (n.terminated ? kDead : 0u);
    }
    static constexpr uint32_t kDead = 0x80000000u;
    bfs_stats stats{};
    std::string error;
    size_t max_nodes = (size_t)1 << 22;
    std::unique_ptr<Pool> pool;      // per-node loops of a batch
    std::unique_ptr<Lane> eval_lane; // value-network calls, beside the host work on the other children
    // buffers of iterate(), kept between calls
    std::vector<float> parent_next_score, eval_scores, move_scores_buf;
    std::vector<uint8_t> flags, parent_more, mv_from, mv_filt, mv_per;
    std::vector<int> to_eval, kept, cand;
    std::vector<unsigned> bin_idx;
    std::vector<uint8_t> live;
    std::vector<int8_t> tt_index;    // of the nodes being scored: index of the table move in their move list, or -1
    std::unique_ptr<Lane> policy_lane;   // the blocking half of the move-scoring call, beside the tree update
    std::vector<Node*> to_score;
    std::vector<Node> fresh;         // the children of the batch in flight; the ones that stay open move to the arena
    std::vector<float> first_score;  // of the nodes score_moves() handled: the score of their best move
    std::vector<size_t> move_base;
    std::vector<EvalRecord> recs, policy_recs;
    size_t grain_min = 64;   // fewest nodes per piece of a parallel loop (BFS_GRAIN in the environment: tuning)
    // pieces of a parallel loop over n nodes: about four per thread, never smaller than grain_min
    // Batches below `wide_batch` nodes run on the pool's narrow group (threads.h): set per batch, the same for
    // every loop of the batch.  (Which threads run a loop never changes what the search does.)
    size_t wide_batch = 320;
    bool wide_now = true;
    size_t grain(size_t n) const {
        const size_t active = (size_t)(wide_now ? pool->threads() : pool->narrow_threads());
        return std::max(grain_min, n / (4 * active) + 1);
    }
    template <class F>
    void par(size_t n, size_t g, F&& f) { pool->for_range(n, g, std::forward<F>(f), wide_now); }

    // ---- transposition table (transposition_table.py) ------------------------------------
    TTEntry& entry(u64 h) { return tt[h & tt_mask]; }

    struct TreeNodeView {   // tt_add's view of an arena node (struct Hot)
        struct {
            u64 hash;
        } b;
        float best_value, separator;
        uint8_t depth;
    };
    void tt_add(const Hot& h, Move following) { tt_add(TreeNodeView{{h.hash}, h.best_value, h.separator, h.depth}, following); }
    void tt_add(const TreeNodeView& n, Move following) {  // add_board_and_move_to_tt, :86-133
        TTEntry& e = entry(n.b.hash);
        const bool lower_side = n.best_value >= n.separator;
        if (e.depth != kNoEntry) {
            if (e.hash == n.b.hash) {
                if (e.depth == n.depth) {
                    if (lower_side) {
                        if (n.best_value > e.lower) {
                            e.lower = n.best_value;
                            e.move = following;
                        }
                    } else if (n.best_value < e.upper) {
                        e.upper = n.best_value;
                    }
                } else if (e.depth < n.depth) {
                    e.depth = n.depth;
                    if (lower_side) {
                        e.move = following;
                        e.upper = FMAX;
                        e.lower = n.best_value;
                    } else {
                        e.upper = n.best_value;
                        e.lower = FMIN;
                    }
                }
            } else {  // always replace
                e.hash = n.b.hash;
                e.depth = n.depth;
                if (lower_side) {
                    e.move = following;
                    e.upper = FMAX;
                    e.lower = n.best_value;
                } else {
                    e.move = kNoMove;
                    e.upper = n.best_value;
                    e.lower = FMIN;
                }
            }
        } else {
            e.hash = n.b.hash;
            e.depth = n.depth;
            if (lower_side) {
                e.move = following;
                e.lower = n.best_value;
            } else {
                e.upper = n.best_value;
            }
        }
    }

    // should_terminate_from_tt, numba_negamax_zero_window.py:203-222
    bool tt_terminates(Node& n) {
        const TTEntry& e = entry(n.b.hash);
        if (e.depth != kNoEntry && e.hash == n.b.hash && e.depth >= n.depth) {
            if (e.lower >= n.separator) {
                n.best_value = e.lower;
                return true;
            }
            if (e.upper < n.separator) {
                n.best_value = e.upper;
                return true;
            }
            if (e.lower > n.best_value) n.best_value = e.lower;
        }
        return false;
    }

    // ---- tree updates (numba_negamax_zero_window.py:399-419) --------------------------------
    void update_from_value(int id, float value, Move following, bool new_termination = true) {
        while (id >= 0) {
            Hot& n = hot[id];
            if (n.terminated) {
                if (!(value > n.best_value)) return;
                n.best_value = value;
                tt_add(n, following);
                value = -value;
                following = n.prev_move;
                id = n.parent;
                new_termination = false;
            } else {
                if (new_termination) n.children_left -= 1;
                n.best_value = std::max(value, n.best_value);
                if (!(n.best_value >= n.separator || n.children_left == 0)) return;
                n.terminated = true;
                up[id] |= kDead;
                tt_add(n, following);
                value = -n.best_value;
                following = n.prev_move;
                id = n.parent;
                new_termination = true;
            }
        }
    }

    bool should_not_terminate(int id) const {  // global_open_priority_nodes.py:4-11
        for (;;) {
            const uint32_t u = up[id];
            if (u & kDead) return false;
            if (u == 0) return true;   // the root
            id = (int)(u - 1);
        }
    }

    // can_draw_from_repetition, numba_negamax_zero_window.py:109-149 (no pre-search game history)
    bool repetition(const Node& child) const {
        if (child.b.halfmove < 4) return false;
        int found = 1;
        if (child.parent < 0 || nodes[child.parent].parent < 0) return false;
        int id = nodes[child.parent].parent;
        if (nodes[id].b.hash == child.b.hash) ++found;
        if (found >= 3) return true;
        for (int num = nodes[id].b.halfmove; num > 1; num -= 2) {
            if (num < 3 && found == 1) return false;
            if (nodes[id].parent < 0 || nodes[nodes[id].parent].parent < 0) return false;
            id = nodes[nodes[id].parent].parent;
            if (nodes[id].b.hash == child.b.hash && ++found >= 3) return true;
        }
        return false;
    }

    static float next_best_move(Node& n) {  // set_up_next_best_move, :152-161
        int best = 0;
        for (int i = 1; i < n.n_moves; ++i)
            if (n.move_scores[i] > n.move_scores[best]) best = i;
        const float s = n.n_moves ? n.move_scores[best] : FMIN;
        if (s == FMIN) n.next_move_index = kNoMoreMoves;
        else {
            n.next_move_index = (uint8_t)best;
            n.move_scores[best] = FMIN;
        }
        return s;
    }

    // Draw / table / mate checks and move generation for a freshly made node
    // (depth_zero_should_terminate_array :227-259, child_termination_check_and_move_gen :284-312).
    // Returns true when the node still needs work: an evaluation (depth 0) or move scores (depth > 0).
    bool settle(Node& n) {
        if (n.b.halfmove >= 50 || insufficient_material(n.b)) {
            n.terminated = true;
            n.best_value = TIE;
            return false;
        }
        if (repetition(n)) {
            n.terminated = true;
            n.best_value = TIE;
            return false;
        }
        if (tt_terminates(n)) {
            n.terminated = true;
            return false;
        }
        if (n.depth == 0) {
            if (has_legal_move(n.b)) return true;
            n.terminated = true;
            n.best_value = in_check(n.b) ? g_loss[0] : TIE;
            return false;
        }
        n.n_moves = (uint8_t)legal_moves(n.b, n.moves);
        n.children_left = n.n_moves;
        if (!n.n_moves) {
            n.terminated = true;
            n.best_value = in_check(n.b) ? g_loss[n.depth] : TIE;
            return false;
        }
        return true;
    }

    // A legal move stored in the table goes first (has_legal_tt_move, :262-281): its index in the node's moves, or -1.
    int table_move_index(const Node& n) const {
        const TTEntry& e = tt[n.b.hash & tt_mask];
        if (e.depth == kNoEntry || e.hash != n.b.hash || e.move.from == 255) return -1;
        for (int i = 0; i < n.n_moves; ++i)
            if (n.moves[i] == e.move) return i;
        return -1;
    }

    // Move scores of the listed nodes, in two halves so that the tree update can run between them while the GPU
    // works (start_move_scoring :164-180, get_move_from_and_filter_squares_and_sizes :485-527; the reference's fake
    // when no network is given).
    // score_begin: packs records and move lists, starts stage 1, hands stage 2 (which blocks until the GPU is done) to
    // the policy lane, and looks up each node's table move NOW, i.e. in the table as it is before this batch's
    // updates, which is when the reference looks (has_legal_tt_move at child creation, :262-281).
    struct FinishCall {
        bfs_search* s;
        size_t boards, total;
    };
    static int run_finish(void* arg) {
        FinishCall* c = static_cast<FinishCall*>(arg);
        bfs_search* s = c->s;
        return s->cfg.move_finish(s->cfg.move_user, s->mv_from.data(), s->mv_filt.data(), s->mv_per.data(), c->boards, c->total,
                                  s->move_scores_buf.data());
    }
    FinishCall finish_call{};
    bool finish_pending = false;
    // stage 1 (copy the records in, launch) also runs on the policy lane: the call takes 20-30 us on whoever makes it,
    // a tenth of a depth-5 iteration, and the caller packs the move lists meanwhile.  The lane runs its calls in
    // order, so the finish call handed over after the packing finds stage 1 started.
    static int run_start(void* arg) {
        bfs_search* s = static_cast<bfs_search*>(arg);
        return s->cfg.move_start(s->cfg.move_user, s->policy_recs.data(), sizeof(EvalRecord), nullptr, s->policy_recs.size());
    }

    int score_begin(const std::vector<Node*>& who) {
        finish_pending = false;
        if (who.empty()) return 0;
        move_base.resize(who.size() + 1);
        size_t total = 0;
        for (size_t i = 0; i < who.size(); ++i) {
            move_base[i] = total;
            total += who[i]->n_moves;
        }
        move_base[who.size()] = total;
        move_scores_buf.resize(total);
        first_score.resize(who.size());
        tt_index.resize(who.size());
        float* scores = move_scores_buf.data();
        if (cfg.move_start && cfg.move_finish) {
            policy_recs.resize(who.size());
            mv_from.resize(total);
            mv_filt.resize(total);
            mv_per.resize(who.size());
            // stage 1 is started on the records alone; the move lists are packed while the GPU runs it
            par(who.size(), 4 * grain(who.size()), [&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) to_record(who[i]->b, &policy_recs[i]);
            });
            ++stats.policy_calls;
            policy_lane->submit(&run_start, this);
            finish_pending = true;   // (the lane has work of this batch: score_end or LanesJoined waits for it)
        } else {
            for (size_t i = 0; i < total; ++i) scores[i] = total > 1 ? (float)((double)i / (double)(total - 1)) : 0.0f;
        }
        const bool net = cfg.move_start && cfg.move_finish;
        par(who.size(), grain(who.size()), [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const Node& n = *who[i];
                tt_index[i] = (int8_t)table_move_index(n);
                if (!net) continue;
                mv_per[i] = n.n_moves;
                const int flip = n.b.turn ? 0 : 56;  // SQUARES_180 for black: the mover's frame
                size_t m = move_base[i];
                for (int k = 0; k < n.n_moves; ++k, ++m) {
                    const int f = n.moves[k].from ^ flip, t = n.moves[k].to ^ flip;
                    mv_from[m] = (uint8_t)f;
                    mv_filt[m] = g_filter[f][t][n.moves[k].promo < 6 ? n.moves[k].promo : 0];
                }
            }
        });
        if (net) {
            finish_call = FinishCall{this, who.size(), total};
            policy_lane->submit(&run_finish, &finish_call);
            finish_pending = true;
        }
        stats.moves_scored += total;
        return 0;
    }

    // score_end: waits for stage 2, hands every node its scores, puts its table move first and picks its first move
    // (first_score[i]).
    int score_end(const std::vector<Node*>& who) {
        if (finish_pending) {
            finish_pending = false;
            const double tw = now();
            const int rc = policy_lane->wait();
            stats.eval_wait_seconds += now() - tw;
            if (rc) return fail("the move-scoring network's start or finish call failed");
        }
        if (who.empty()) return 0;
        const float* scores = move_scores_buf.data();
        par(who.size(), grain(who.size()), [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                Node& n = *who[i];
                const float* src = scores + move_base[i];
                for (int k = 0; k < n.n_moves; ++k) n.move_scores[k] = src[k];
                if (tt_index[i] >= 0) n.move_scores[tt_index[i]] = g_tt_move_score;
                first_score[i] = next_best_move(n);   // the node is ready for the open list
            }
        });
        return 0;
    }

    // ---- open list (global_open_priority_nodes.py) ------------------------------------------
    size_t bin_of(float score) const {
        const float x = std::fabs(score - cfg.bin_zero_shift);
        // np.digitize on decreasing edges: the number of edges strictly greater than x
        size_t lo = 0, hi = bins_desc.size();
        while (lo < hi) {
            size_t mid = (lo + hi) / 2;
            if (bins_desc[mid] > x) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }

    void clear_bins() {
        for (auto& b : bins) b.v.clear();
        in_bins = 0;
    }

    // Appends to `batch` those of cand[from..) that are still worth expanding, in order; the checks (a walk up
    // the ancestors, read-only) run on the pool.
    void keep_live(const std::vector<int>& cand_ids, size_t from, std::vector<int>* batch) {
        const double tl = now();
        const size_t n = cand_ids.size() - from;
        live.resize(n);
        par(n, grain(n), [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) live[i] = should_not_terminate(cand_ids[from + i]) ? 1 : 0;
        });
        const double tm = now();
        for (size_t i = 0; i < n; ++i)
            if (live[i]) batch->push_back(cand_ids[from + i]);
        stats.select_pop_seconds += now() - tm;
        stats.select_live_seconds += tm - tl;
    }

    // insert_nodes + get_batch (global_open_priority_nodes.py:73-121,143-177).  Nodes leave the bins in the same order
    // as one at a time: candidates are popped in rounds of exactly as many as the batch still lacks, so the last one
    // popped is the one that fills it.
    // Two phases, because the tree update of the batch before may still run on the evaluation lane while the new nodes
    // are put into the bins (phase one touches ids, scores, bins and cand only); the walks up the ancestry in phase two
    // need the update finished (zero_window() joins it between the two).
    bool small_round = false;
    void insert_nodes(const std::vector<int>& ids, const std::vector<float>& scores) {
        stats.nodes_in += ids.size();
        wide_now = ids.size() >= wide_batch;
        cand.clear();
        small_round = ids.size() + in_bins < cfg.max_batch_size;
        if (small_round) {
            for (auto& bin : bins) {
                auto& b = bin.v;
                cand.insert(cand.end(), b.rbegin(), b.rend());
                b.clear();
            }
            in_bins = 0;
            cand.insert(cand.end(), ids.begin(), ids.end());
        } else {
            const double ti = now();
            bin_idx.resize(ids.size());
            par(ids.size(), 4 * grain(ids.size()), [&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) bin_idx[i] = (unsigned)bin_of(scores[i]);
            });
            for (size_t i = 0; i < ids.size(); ++i) bins[bin_idx[i]].v.push_back(ids[i]);
            in_bins += ids.size();
            stats.select_insert_seconds += now() - ti;
        }
    }
    std::vector<int> get_batch() {
        std::vector<int> batch;
        if (small_round) {
            keep_live(cand, 0, &batch);
        } else {
            size_t bin = 0;
            while (batch.size() < cfg.max_batch_size && in_bins > 0) {
                const double tp = now();
                const size_t from = cand.size();
                size_t need = cfg.max_batch_size - batch.size();
                while (need > 0 && bin < bins.size()) {
                    auto& b = bins[bin].v;
                    if (b.empty()) {
                        ++bin;
                        continue;
                    }
                    // as many as are needed off the back of the bin, last one first (= one pop at a time)
                    const size_t k = std::min(need, b.size());
                    cand.insert(cand.end(), b.rbegin(), b.rbegin() + (std::ptrdiff_t)k);
                    b.resize(b.size() - k);
                    in_bins -= k;
                    need -= k;
                }
                stats.select_pop_seconds += now() - tp;
                if (cand.size() == from) break;
                keep_live(cand, from, &batch);
            }
        }
        stats.nodes_out += batch.size();
        return batch;
    }

    // ---- one batch (do_iteration, :572-693) ------------------------------------------------
    struct EvalCall {
        bfs_search* s;
        size_t n;
    };
    static int run_eval(void* arg) {
        EvalCall* c = static_cast<EvalCall*>(arg);
        size_t done = 0;
        const int rc = c->s->cfg.board_eval(c->s->cfg.eval_user, c->s->recs.data(), sizeof(EvalRecord), nullptr, c->n,
                                            c->s->eval_scores.data(), &done);
        return rc ? rc : (done != c->n ? -1 : 0);
    }

    // update_tree_from_terminating_nodes, :422-453: the batch's children in index order, one thread.  Reads the
    // children's hash, parent, move, termination and value; writes arena nodes (their ancestors) and the table.
    static constexpr int kOverlapUpdateThreads = 4;
    struct UpdateCall {
        bfs_search* s;
        size_t n;
    };
    UpdateCall update_call{};
    bool update_pending = false;   // the batch's tree update runs on the evaluation lane (iterate() left it there)
    void finish_update() {         // what of the update is left when everything beside it is done
        if (!update_pending) return;
        update_pending = false;
        const double t0 = now();
        (void)eval_lane->wait();
        stats.update_seconds += now() - t0;
    }
    static int run_update(void* arg) {
        UpdateCall* c = static_cast<UpdateCall*>(arg);
        const double t0 = now();
        c->s->update_tree(c->n);
        c->s->stats.update_busy_seconds += now() - t0;
        return 0;
    }
    void update_tree(size_t n) {
        size_t e = 0;
        for (size_t j = 0; j < n; ++j) {
            Node& c = fresh[j];
            if (j + 16 < n) __builtin_prefetch(&hot[fresh[j + 16].parent], 1);   // the parent's tree state, then ...
            if (j + 8 < n) {   // ... the table lines this loop will touch a few children from now: the child's and its parent's
                const Node& ahead = fresh[j + 8];
                __builtin_prefetch(&entry(ahead.b.hash), 1);
                __builtin_prefetch(&entry(hot[ahead.parent].hash), 1);
            }
            if (e < to_eval.size() && to_eval[e] == (int)j) {
                const float s = eval_scores[e++] - cfg.eval_zero_shift;
                TTEntry& t = entry(c.b.hash);  // add_evaluated_boards_to_tt, transposition_table.py:138-148
                if (t.depth == kNoEntry) {
                    t.hash = c.b.hash;
                    t.depth = 0;
                    t.lower = t.upper = s;
                }
                update_from_value(c.parent, -s, c.prev_move);
            } else if (c.terminated) {
                update_from_value(c.parent, -c.best_value, c.prev_move);
            }
        }
    }

    // The per-node loops run on the pool; everything that orders nodes (which boards go to the evaluators and
    // in which order, tree and table updates, the open list) stays on the calling thread and walks the batch
    // in index order, so a search does the same thing whatever the number of threads.
    // The lanes are handed pointers into iterate()'s frame and into the search's per-batch buffers: however the
    // function is left -- a return, or std::bad_alloc out of any buffer that grows between a submit and its wait --
    // nothing of the batch may still be running.  (Lane::wait is cheap and idempotent when the lane is idle.)
    struct LanesJoined {
        bfs_search* s;
        bool leave_update = false;   // a successful iterate() hands its running update to zero_window()
        ~LanesJoined() {
            if (!leave_update) {
                (void)s->eval_lane->wait();
                s->update_pending = false;
            }
            (void)s->policy_lane->wait();
            s->finish_pending = false;
        }
    };

    int iterate(const std::vector<int>& batch, std::vector<int>* to_insert, std::vector<float>* insert_scores) {
        LanesJoined joined{this};
        ++stats.iterations;
        const size_t n = batch.size();
        wide_now = n >= wide_batch;
        if (wide_now) ++stats.wide_iterations;
        double t0 = now(), t1;
        parent_next_score.resize(n);
        parent_more.resize(n);
        flags.resize(n);
        // Children are made in a block of their own that every batch reuses (it stays in cache); most of them end
        // at birth (depth 0, draws, table hits, mates), and only the ones that stay open are copied to the end of
        // the arena afterwards: the arena holds live nodes only and nothing in it ever moves.
        if (fresh.size() < n) fresh.resize(n);
        // depth-zero children first, so that the evaluator gets its boards as early as possible
        par(n, grain(n), [&](size_t b, size_t e) {
            for (size_t j = b; j < e; ++j) {
                if (j + 6 < e) {   // parents lie anywhere in an arena of gigabytes: ask for their lines a few nodes ahead
                    const char* q = reinterpret_cast<const char*>(&nodes[batch[j + 6]]);
                    for (size_t off : {(size_t)0, (size_t)64, offsetof(Node, moves), offsetof(Node, moves) + 64,
                                       offsetof(Node, move_scores), offsetof(Node, move_scores) + 64, offsetof(Node, move_scores) + 128})
                        __builtin_prefetch(q + off, 1);
                }
                Node& p = nodes[batch[j]];
                Node& c = fresh[j];
                const Move m = p.moves[p.next_move_index];
                parent_next_score[j] = next_best_move(p);
                parent_more[j] = p.next_move_index != kNoMoreMoves;   // (nothing below changes it: kept for the re-insertion list)
                c.b = p.b;
                push(c.b, m);
                c.parent = batch[j];
                c.separator = -p.separator;
                c.best_value = FMIN;
                c.depth = (uint8_t)(p.depth - 1);
                c.terminated = false;
                c.n_moves = 0;
                c.next_move_index = kNoMoreMoves;
                c.children_left = 0;
                c.prev_move = m;
            }
            // second pass over the piece: the checks of the depth-0 children, their table lines asked for ahead
            for (size_t j = b; j < e; ++j) {
                if (j + 6 < e) __builtin_prefetch(&tt[fresh[j + 6].b.hash & tt_mask]);
                Node& c = fresh[j];
                flags[j] = (c.depth == 0 && settle(c)) ? 1 : 0;
            }
        });
        to_eval.clear();
        for (size_t j = 0; j < n; ++j)
            if (flags[j]) to_eval.push_back((int)j);
        EvalCall call{this, to_eval.size()};
        if (!to_eval.empty()) {
            recs.resize(to_eval.size());
            eval_scores.resize(to_eval.size());
            par(to_eval.size(), 4 * grain(to_eval.size()), [&](size_t b, size_t e) {
                for (size_t k = b; k < e; ++k) to_record(fresh[to_eval[k]].b, &recs[k]);
            });
            stats.boards_evaluated += to_eval.size();
            ++stats.eval_calls;
            stats.max_batch = std::max<uint64_t>(stats.max_batch, to_eval.size());
            eval_lane->submit(&run_eval, &call);
        }
        t1 = now(); stats.expand_seconds += t1 - t0; t0 = t1;
        // meanwhile: the other children (draws, table hits, mates, move generation) and their move scores
        par(n, grain(n), [&](size_t b, size_t e) {
            for (size_t j = b; j < e; ++j) {
                if (j + 6 < e) __builtin_prefetch(&tt[fresh[j + 6].b.hash & tt_mask]);
                Node& c = fresh[j];
                if (c.depth != 0) flags[j] = settle(c) ? 2 : 0;
            }
        });
        to_score.clear();
        kept.clear();
        for (size_t j = 0; j < n; ++j)
            if (flags[j] == 2) {
                to_score.push_back(&fresh[j]);
                kept.push_back((int)j);
            }
        t1 = now(); stats.settle_seconds += t1 - t0; t0 = t1;
        const int score_rc = score_begin(to_score);
        t1 = now(); stats.policy_seconds += t1 - t0; t0 = t1;
        const int eval_rc = to_eval.empty() ? 0 : eval_lane->wait();
        t1 = now(); stats.value_wait_seconds += t1 - t0;
        stats.eval_wait_seconds += t1 - t0;
        t0 = t1;
        if (eval_rc || score_rc) return eval_rc ? fail("the value network's evaluation call failed") : score_rc;
        // The tree update (one thread, in the batch's order) runs on the evaluation lane, which is idle now, while
        // this thread and the pool give the open children their move scores and move them to the arena: the update
        // reads the children's hash / parent / move / value and writes their ANCESTORS and the table; the rest of
        // this function writes the children's move lists and new arena slots and reads the parents' next_move_index,
        // which the update never touches.
        // With few host threads the overlap loses: the update then leaves the parents' and the table's lines in
        // another core's cache for the one thread that expands the next batch (depth 5, one thread: 3.3 -> 2.8 M
        // nodes/s; twelve threads: 7.3 -> 7.8 M), so it runs here.
        update_call = UpdateCall{this, n};
        const bool beside = wide_now && pool->threads() >= kOverlapUpdateThreads;
        if (beside) {
            ++stats.overlapped_updates;
            eval_lane->submit(&run_update, &update_call);
            update_pending = true;
        } else {
            (void)run_update(&update_call);
        }
        if (!beside) { t1 = now(); stats.update_seconds += t1 - t0; t0 = t1; }
        if (int rc = score_end(to_score)) return rc;
        t1 = now(); stats.policy_seconds += t1 - t0; t0 = t1;
        // parents with moves left go back to the open list, open children join it
        to_insert->clear();
        insert_scores->clear();
        for (size_t j = 0; j < n; ++j)
            if (parent_more[j]) {   // = nodes[batch[j]].next_move_index != kNoMoreMoves, without going back to the arena
                to_insert->push_back(batch[j]);
                insert_scores->push_back(parent_next_score[j]);
            }
        // open children (exactly the ones whose moves were scored, in that order) move to the arena and join the list
        const size_t base = nodes.size();
        if (base + kept.size() > max_nodes)
            return fail("search arena would exceed its limit of %zu nodes (bfs_config.max_nodes)", max_nodes);
        nodes.resize(base + kept.size());   // may throw; LanesJoined joins the update before the exception leaves
        stats.arena_peak = std::max<uint64_t>(stats.arena_peak, nodes.size());
        const double tc = now();
        par(kept.size(), grain(kept.size()), [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const Node& src = fresh[kept[i]];
                Node& dst = nodes[base + i];
                if (i + 8 < e) {   // the end of the arena is cold memory: ask for the lines a few nodes ahead
                    const char* p = reinterpret_cast<const char*>(&nodes[base + i + 8]);
                    for (size_t off : {(size_t)0, (size_t)64, offsetof(Node, moves), offsetof(Node, moves) + 64,
                                       offsetof(Node, move_scores), offsetof(Node, move_scores) + 64, offsetof(Node, move_scores) + 128})
                        __builtin_prefetch(p + off, 1);
                }
                std::memcpy(static_cast<void*>(&dst), static_cast<const void*>(&src), offsetof(Node, moves));
                enter_tree((int)(base + i), src);   // (open children are not terminated)
                std::memcpy(dst.moves, src.moves, src.n_moves * sizeof(Move));
                std::memcpy(dst.move_scores, src.move_scores, src.n_moves * sizeof(float));
            }
        });
        stats.compact_copy_seconds += now() - tc;
        for (size_t i = 0; i < kept.size(); ++i) {
            to_insert->push_back((int)(base + i));
            insert_scores->push_back(first_score[i]);
        }
        t1 = now(); stats.compact_seconds += t1 - t0;
        joined.leave_update = true;   // zero_window() joins it (finish_update) after the new nodes are in the bins
        return 0;
    }

    // set_up_root_node_for_struct, :715-767
    int make_root(const Board& start, int depth, float separator) {
        nodes.clear();
        clear_bins();
        nodes.resize(1);
        Node& r = nodes[0];
        r.b = start;
        to_whites_perspective(r.b);
        r.parent = -1;
        r.separator = separator;
        r.best_value = FMIN;
        r.depth = (uint8_t)depth;
        r.terminated = false;
        r.n_moves = 0;
        r.next_move_index = kNoMoreMoves;
        r.children_left = 0;
        r.prev_move = kNoMove;
        struct RootEnters {   // however this function is left, the root's tree state is what the node holds by then
            bfs_search* s;
            ~RootEnters() { s->enter_tree(0, s->nodes[0]); }
        } root_enters{this};
        if (depth == 0) {
            // the reference never searches to depth 0; give the evaluation so callers get a value
            if (settle(r)) {
                EvalRecord rec;
                to_record(r.b, &rec);
                float s;
                size_t done = 0;
                if (cfg.board_eval(cfg.eval_user, &rec, sizeof rec, nullptr, 1, &s, &done) || done != 1)
                    return fail("the value network's evaluation call failed");
                ++stats.boards_evaluated;
                r.best_value = s - cfg.eval_zero_shift;
                r.terminated = true;
            }
            return 0;
        }
        if (!settle(r)) return 0;
        LanesJoined joined{this};
        const double t0 = now();
        to_score.assign(1, &nodes[0]);
        wide_now = false;
        if (int rc = score_begin(to_score)) return rc;
        if (int rc = score_end(to_score)) return rc;
        stats.policy_seconds += now() - t0;
        return 0;
    }

    // zero_window_negamax_search, :696-712
    int zero_window(const Board& start, int depth, float separator, float* value) {
        const double t0 = now();
        if (int rc = make_root(start, depth, separator)) return rc;
        if (!hot[0].terminated) {
            std::vector<int> batch{0}, to_insert;
            std::vector<float> scores;
            stats.nodes_out += 1;
            struct UpdateJoined {   // nothing of a batch runs on once this function is left, however it is left
                bfs_search* s;
                ~UpdateJoined() { s->finish_update(); }
            } update_joined{this};
            while (!batch.empty()) {
                if (int rc = iterate(batch, &to_insert, &scores)) return rc;
                // the batch's tree update may still run (evaluation lane): the new nodes go into the bins beside it
                const bool nothing_open = to_insert.empty() && in_bins == 0;
                double ts = now();
                if (!nothing_open) insert_nodes(to_insert, scores);
                stats.select_seconds += now() - ts;
                finish_update();
                if (hot[0].terminated) {
                    clear_bins();
                    break;
                }
                if (nothing_open) break;
                ts = now();
                batch = get_batch();
                stats.select_seconds += now() - ts;
            }
        }
        *value = hot[0].best_value;
        stats.seconds += now() - t0;
        return 0;
    }

    // mtd_f, :778-830
    int mtd_f(const Board& start, int depth, float first_guess, float inc, float* value, Move* move) {
        float guess = first_guess, upper = g_win[0], lower = g_loss[0];
        u64 root_hash = 0;
        while (lower < upper) {
            float beta;
            if (lower == g_loss[0]) {
                if (upper != g_win[0]) beta = std::min(upper - inc, std::nextafterf(upper, FMIN));
                else beta = guess;
            } else if (upper == g_win[0]) {
                beta = std::max(lower + inc, std::nextafterf(lower, FMAX));
            } else {
                beta = std::max(lower + (upper - lower) / 2, std::nextafterf(lower, FMAX));
            }
            const float sep = std::nextafterf(beta, FMIN);
            if (int rc = zero_window(start, depth, sep, &guess)) return rc;
            root_hash = nodes[0].b.hash;
            if (guess < beta) upper = guess;
            else lower = guess;
        }
        *value = guess;
        Move m = entry(root_hash).hash == root_hash ? entry(root_hash).move : kNoMove;  // choose_move, transposition_table.py:33-48
        if (m.from != 255 && !start.turn) {
            m.from ^= 56;
            m.to ^= 56;
        }
        *move = m;
        return 0;
    }
};

// ---- C ABI -------------------------------------------------------------------------------------

extern "C" const char* bfs_last_error(void) { return g_err; }

extern "C" int bfs_perft(const char* fen, int depth, uint64_t* out) {
    Board b;
    if (!fen || !out || !board_from_fen(fen, &b)) return fail("bfs_perft: bad FEN");
    if (depth < 0 || depth > 12) return fail("bfs_perft: depth %d out of range", depth);
    *out = perft(b, depth);
    return 0;
}

extern "C" int bfs_legal_moves(const char* fen, uint8_t* moves, int* n_moves) {
    Board b;
    if (!fen || !moves || !n_moves || !board_from_fen(fen, &b)) return fail("bfs_legal_moves: bad argument");
    Move mv[kMaxMoves];
    const int n = legal_moves(b, mv);
    for (int i = 0; i < n; ++i) {
        moves[3 * i] = mv[i].from;
        moves[3 * i + 1] = mv[i].to;
        moves[3 * i + 2] = mv[i].promo;
    }
    *n_moves = n;
    return 0;
}

// Legal moves of a node record (bytes 0..81; the move list of a reference record is ignored).
extern "C" int bfs_legal_moves_record(const void* record, uint8_t* moves, int* n_moves) {
    if (!record || !moves || !n_moves) return fail("bfs_legal_moves_record: null pointer");
    init_tables();
    EvalRecord r;
    std::memcpy(&r, record, 82);
    Board b;
    from_record(r, &b);
    Move mv[kMaxMoves];
    const int n = legal_moves(b, mv);
    for (int i = 0; i < n; ++i) {
        moves[3 * i] = mv[i].from;
        moves[3 * i + 1] = mv[i].to;
        moves[3 * i + 2] = mv[i].promo;
    }
    *n_moves = n;
    return 0;
}

extern "C" int bfs_push(const char* fen, const uint8_t* move, char* fen_out, size_t cap) {
    Board b;
    if (!fen || !move || !fen_out || !board_from_fen(fen, &b)) return fail("bfs_push: bad argument");
    const Move m{move[0], move[1], move[2]};
    if (!is_legal(b, m)) return fail("bfs_push: illegal move %u %u %u", move[0], move[1], move[2]);
    push(b, m);
    if (b.hash != compute_hash(b)) return fail("bfs_push: incremental hash differs from the recomputed one");
    const std::string s = board_to_fen(b);
    if (s.size() + 1 > cap) return fail("bfs_push: output buffer too small");
    std::memcpy(fen_out, s.c_str(), s.size() + 1);
    return 0;
}

extern "C" int bfs_is_legal(const char* fen, const uint8_t* moves, size_t n, uint8_t* by_generator, uint8_t* direct) {
    Board b;
    if (!fen || !moves || !by_generator || !direct || !board_from_fen(fen, &b)) return fail("bfs_is_legal: bad argument");
    for (size_t i = 0; i < n; ++i) {
        const Move m{moves[3 * i], moves[3 * i + 1], moves[3 * i + 2]};
        by_generator[i] = is_legal(b, m) ? 1 : 0;
        direct[i] = is_legal_direct(b, m) ? 1 : 0;
    }
    return 0;
}

extern "C" int bfs_push_record(const void* record, const uint8_t* move, void* record88_out) {
    if (!record || !move || !record88_out) return fail("bfs_push_record: null pointer");
    init_tables();
    EvalRecord r;
    std::memcpy(&r, record, 82);
    Board b;
    from_record(r, &b);
    const Move m{move[0], move[1], move[2]};
    if (!is_legal(b, m)) return fail("bfs_push_record: illegal move %u %u %u", move[0], move[1], move[2]);
    push(b, m);
    to_record(b, static_cast<EvalRecord*>(record88_out));
    return 0;
}

extern "C" int bfs_hash_walk(const char* fen, int lines, int plies, uint64_t seed, uint64_t* positions_checked,
                             uint64_t* mismatches) {
    Board start;
    if (!fen || !positions_checked || !mismatches || !board_from_fen(fen, &start)) return fail("bfs_hash_walk: bad argument");
    u64 x = seed * 0x9E3779B97F4A7C15ull + 1, checked = 0, bad = 0;
    for (int l = 0; l < lines; ++l) {
        Board b = start;
        for (int p = 0; p < plies; ++p) {
            Move mv[kMaxMoves];
            const int k = legal_moves(b, mv);
            if (!k) break;
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            push(b, mv[x % (u64)k]);   // the hash travels with the board, updated move by move
            ++checked;
            if (b.hash != compute_hash(b)) ++bad;
        }
    }
    *positions_checked = checked;
    *mismatches = bad;
    return 0;
}

extern "C" int bfs_record_from_fen(const char* fen, void* record88) {
    Board b;
    if (!fen || !record88 || !board_from_fen(fen, &b)) return fail("bfs_record_from_fen: bad argument");
    to_record(b, static_cast<EvalRecord*>(record88));
    return 0;
}

// Random legal playout: the position after every ply, as records (legal positions for the
// evaluation tests: SURVEY.md 8(d) "tier-2 generator").
extern "C" int bfs_random_playout_records(const char* fen, int plies, uint64_t seed, void* records88, int* n_records) {
    Board b;
    if (!fen || !records88 || !n_records || !board_from_fen(fen, &b)) return fail("bfs_random_playout_records: bad argument");
    EvalRecord* out = static_cast<EvalRecord*>(records88);
    u64 x = seed * 0x9E3779B97F4A7C15ull + 1;
    int n = 0;
    for (int p = 0; p < plies; ++p) {
        Move mv[kMaxMoves];
        const int k = legal_moves(b, mv);
        if (!k || b.halfmove >= 100 || insufficient_material(b)) break;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        push(b, mv[x % (u64)k]);
        to_record(b, &out[n++]);
    }
    *n_records = n;
    return 0;
}

// bf_piece_sum_eval, code_testing.py:516-532: material in the mover's frame, the evaluator the
// reference tests its search with.  Values per piece type P N B R Q K = 100 300 300 500 900 0.
extern "C" int bfs_material_eval(void*, const void* records, size_t stride, const uint8_t* mask, size_t n, float* scores,
                                 size_t* n_scored) {
    const unsigned char* p = static_cast<const unsigned char*>(records);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        EvalRecord r;
        std::memcpy(&r, p + i * stride, 82);
        const u64 own = r.turn ? r.occ_white : r.occ_black, opp = r.turn ? r.occ_black : r.occ_white;
        auto side = [&](u64 s) {
            return 100 * __builtin_popcountll(r.pawns & s) + 300 * __builtin_popcountll((r.knights | r.bishops) & s) +
                   500 * __builtin_popcountll(r.rooks & s) + 900 * __builtin_popcountll(r.queens & s);
        };
        scores[k++] = (float)(side(own) - side(opp));
    }
    if (n_scored) *n_scored = k;
    return 0;
}

namespace {
// create_negamax_function, code_testing.py:548-595: plain depth-limited negamax with the same
// terminal rules as the search; the oracle of the zero-window property test.
int negamax(const Board& b, int depth, float alpha, float beta, bfs_board_eval_fn eval, void* user, float* out) {
    if (b.halfmove >= 50 || insufficient_material(b)) {
        *out = TIE;
        return 0;
    }
    Move mv[kMaxMoves];
    const int n = legal_moves(b, mv);
    if (!n) {
        *out = in_check(b) ? g_loss[depth] : TIE;
        return 0;
    }
    if (depth == 0) {
        EvalRecord r;
        to_record(b, &r);
        size_t done = 0;
        if (eval(user, &r, sizeof r, nullptr, 1, out, &done) || done != 1) return -1;
        return 0;
    }
    float best = FMIN;
    for (int i = 0; i < n; ++i) {
        Board c = b;
        push(c, mv[i]);
        float v;
        if (negamax(c, depth - 1, -beta, -alpha, eval, user, &v)) return -1;
        best = std::max(best, -v);
        alpha = std::max(alpha, best);
        if (alpha >= beta) break;
    }
    *out = best;
    return 0;
}
}  // namespace

extern "C" int bfs_negamax(const char* fen, int depth, bfs_board_eval_fn eval, void* user, float* value) {
    std::call_once(g_const_once, build_constants);
    Board b;
    if (!fen || !eval || !value || !board_from_fen(fen, &b)) return fail("bfs_negamax: bad argument");
    if (depth < 0 || depth >= kMaxDepth) return fail("bfs_negamax: depth %d out of range", depth);
    to_whites_perspective(b);
    if (negamax(b, depth, FMIN, FMAX, eval, user, value)) return fail("bfs_negamax: the evaluation call failed");
    return 0;
}

extern "C" int bfs_search_create(bfs_search** out, const bfs_config* cfg) {
    std::call_once(g_const_once, build_constants);
    if (!out || !cfg) return fail("bfs_search_create: null pointer");
    *out = nullptr;
    if (!cfg->board_eval) return fail("bfs_search_create: no evaluation function");
    if ((cfg->move_start == nullptr) != (cfg->move_finish == nullptr)) return fail("bfs_search_create: give both move-scoring calls or neither");
    if (cfg->max_batch_size == 0) return fail("bfs_search_create: max_batch_size is 0");
    if (cfg->tt_bits < 4 || cfg->tt_bits > 30) return fail("bfs_search_create: tt_bits %d out of range 4..30", cfg->tt_bits);
    if (!cfg->bin_edges || cfg->n_bin_edges == 0) return fail("bfs_search_create: no bin edges");
    bfs_search* s = new (std::nothrow) bfs_search;
    if (!s) return fail("bfs_search_create: out of memory");
    s->cfg = *cfg;
    s->bins_desc.assign(cfg->bin_edges, cfg->bin_edges + cfg->n_bin_edges);
    std::sort(s->bins_desc.begin(), s->bins_desc.end(), [](float a, float b) { return a > b; });
    s->cfg.bin_edges = nullptr;
    s->bins.resize(s->bins_desc.size() + 1);
    if (cfg->threads < 0 || cfg->threads > 256) {
        delete s;
        return fail("bfs_search_create: threads %d out of range 0..256", cfg->threads);
    }
    int threads = cfg->threads;
    if (threads == 0) {
        // Beside the pool run the caller's thread (part of it), the two evaluator lanes and the GPU runtime's own
        // threads: with every core given to the pool the lanes lose theirs (depth 5 on 16 cores: 12 threads 7.8 M
        // nodes/s, 15 threads 5.7 M, profiles/r2/search).  Cores: OMP_NUM_THREADS where a scheduler sets it to the
        // job's share of the machine (hardware_concurrency reports the whole machine there), else what the machine has.
        unsigned cores = std::max(1u, std::thread::hardware_concurrency());
        if (const char* e = std::getenv("OMP_NUM_THREADS")) {
            const unsigned long v = std::strtoul(e, nullptr, 10);
            if (v >= 1 && v < cores) cores = (unsigned)v;
        }
        threads = (int)std::min(16u, cores > 5 ? cores - 4 : std::max(1u, cores / 2));
    }
    if (cfg->max_nodes > (size_t)INT_MAX) {   // node ids (parent links, the open list) are ints
        delete s;
        return fail("bfs_search_create: max_nodes %zu exceeds the limit of %d", cfg->max_nodes, INT_MAX);
    }
    if (cfg->max_nodes) s->max_nodes = cfg->max_nodes;
    try {
        s->tt.assign((size_t)1 << cfg->tt_bits, TTEntry());
        s->nodes.reserve_blocks(s->max_nodes);
        s->up.reset(static_cast<uint32_t*>(std::calloc(s->max_nodes + 1, sizeof(uint32_t))));
        s->hot.reset(static_cast<Hot*>(std::calloc(s->max_nodes + 1, sizeof(Hot))));
        if (!s->up || !s->hot) throw std::bad_alloc();
        s->pool.reset(new Pool(threads));       // runs with the workers it could start
        s->eval_lane.reset(new Lane());
        s->policy_lane.reset(new Lane());
    } catch (const std::exception& e) {
        delete s;
        return fail("bfs_search_create: %s", e.what());
    }
    if (!s->eval_lane->ok() || !s->policy_lane->ok()) {
        delete s;
        return fail("bfs_search_create: could not start the evaluation threads");
    }
    s->cfg.threads = s->pool->threads();
    if (const char* g = std::getenv("BFS_GRAIN")) s->grain_min = std::max<size_t>(1, std::strtoull(g, nullptr, 10));
    if (const char* w = std::getenv("BFS_WIDE_BATCH")) s->wide_batch = std::strtoull(w, nullptr, 10);   // 0: every batch on all threads
    s->tt_mask = ((u64)1 << cfg->tt_bits) - 1;
    *out = s;
    return 0;
}

extern "C" int bfs_search_threads(const bfs_search* s) { return s ? s->pool->threads() : 0; }

namespace {
// the arena, the open list and the evaluators' buffers grow inside a search: out of memory becomes an error code
template <class F>
int guarded(const char* what, F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail("%s: out of memory", what);
    } catch (const std::exception& e) {
        return fail("%s: %s", what, e.what());
    }
}
}  // namespace

extern "C" void bfs_search_destroy(bfs_search* s) { delete s; }

extern "C" void bfs_search_clear_tt(bfs_search* s) {
    if (s) std::fill(s->tt.begin(), s->tt.end(), TTEntry());
}

extern "C" void bfs_search_stats(const bfs_search* s, bfs_stats* out, int reset) {
    if (!s || !out) return;
    *out = s->stats;
    if (reset) const_cast<bfs_search*>(s)->stats = bfs_stats{};
}

extern "C" int bfs_zero_window(bfs_search* s, const char* fen, int depth, float separator, float* value) {
    Board b;
    if (!s || !fen || !value || !board_from_fen(fen, &b)) return fail("bfs_zero_window: bad argument");
    if (depth < 0 || depth >= kMaxDepth) return fail("bfs_zero_window: depth %d out of range", depth);
    // code_testing.py:606: the separator handed to the tree is the float just below the one asked for
    return guarded("bfs_zero_window", [&] { return s->zero_window(b, depth, std::nextafterf(separator, FMIN), value); });
}

extern "C" int bfs_mtd_f(bfs_search* s, const char* fen, int depth, float first_guess, float guess_increment, float* value,
                         uint8_t* move) {
    Board b;
    if (!s || !fen || !value || !move || !board_from_fen(fen, &b)) return fail("bfs_mtd_f: bad argument");
    if (depth < 1 || depth >= kMaxDepth) return fail("bfs_mtd_f: depth %d out of range", depth);
    Move m;
    if (int rc = guarded("bfs_mtd_f", [&] { return s->mtd_f(b, depth, first_guess, guess_increment, value, &m); })) return rc;
    move[0] = m.from; move[1] = m.to; move[2] = m.promo;
    return 0;
}

// iterative_deepening_mtd_f, :833-870 (guess increment .05 per depth, the reference's default)
extern "C" int bfs_iterative_deepening(bfs_search* s, const char* fen, const int* depths, int n_depths, float* value,
                                       uint8_t* move) {
    Board b;
    if (!s || !fen || !depths || n_depths <= 0 || !value || !move || !board_from_fen(fen, &b))
        return fail("bfs_iterative_deepening: bad argument");
    float guess = 0.0f;
    Move m = kNoMove;
    for (int i = 0; i < n_depths; ++i) {
        if (depths[i] < 1 || depths[i] >= kMaxDepth) return fail("bfs_iterative_deepening: depth %d out of range", depths[i]);
        if (int rc = guarded("bfs_iterative_deepening", [&] { return s->mtd_f(b, depths[i], guess, 0.05f, &guess, &m); })) return rc;
    }
    *value = guess;
    move[0] = m.from; move[1] = m.to; move[2] = m.promo;
    return 0;
}
