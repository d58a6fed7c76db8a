// Batched k-best-first zero-window negamax, mtd(f) and iterative deepening: the native search
// driver that feeds the GPU evaluators (BASELINE configs[4]).
//
// Restates batch_first/numba_negamax_zero_window.py (do_iteration :572-693,
// zero_window_negamax_search :696-712, set_up_root_node_for_struct :715-767, mtd_f :778-830,
// iterative_deepening_mtd_f :833-870, update_node_from_value :399-419,
// update_tree_from_terminating_nodes :422-453, should_terminate_from_tt :203-222,
// can_draw_from_repetition :109-149, set_up_next_best_move :152-161),
// batch_first/global_open_priority_nodes.py (PriorityBins :181-240, insert_nodes :73-91,
// get_batch :94-121, insert_and_get_batch :143-177) and batch_first/transposition_table.py
// (add_board_and_move_to_tt :86-133, add_evaluated_boards_to_tt :138-148, choose_move :33-48).
//
// Deliberate differences (search results are unaffected; the reference's own check is a property,
// zero-window result versus full negamax, code_testing.py:435-496):
//   * nodes live in an index-addressed arena instead of Numba jitclass linked lists;
//   * a child's legal moves are generated when the child is created, and a legal move stored in the
//     transposition table is put first with the score TT_MOVE_SCORE_VALUE; the reference delays
//     generation for such nodes (NEXT_MOVE_IS_FROM_TT_VAL, :262-281, :350-363) to save host time;
//   * Zobrist keys differ in value (board.h).
#pragma once
#include <cstddef>
#include <cstdint>

extern "C" {

// scores[i] for the i-th selected record; same signature as bf_eval_structs_host
// (include/batchfirst.h), so the GPU library's entry point can be plugged in directly.
typedef int (*bfs_board_eval_fn)(void* user, const void* records, size_t stride, const uint8_t* mask, size_t n,
                                 float* scores, size_t* n_scored);
// stage 1 / stage 2 of the move-scoring network: bf_policy_start_structs_host / bf_policy_finish_host
typedef int (*bfs_move_start_fn)(void* user, const void* records, size_t stride, const uint8_t* mask, size_t n);
typedef int (*bfs_move_finish_fn)(void* user, const uint8_t* from_squares, const uint8_t* move_filters,
                                  const uint8_t* moves_per_board, size_t n_boards, size_t n_moves, float* scores);

typedef struct {
    size_t max_batch_size;      // PriorityBins.max_batch_size_to_accept (engine.py:214: 5000)
    int tt_bits;                // log2 of the transposition table's entry count (reference: 30)
    const float* bin_edges;     // ascending, as given to PriorityBins (np.linspace(-4000, 4000, 1000) in the tests)
    size_t n_bin_edges;
    float bin_zero_shift;       // PriorityBins.zero_shift
    float eval_zero_shift;      // engine.py:260: scores - zero_shift
    void* eval_user;
    bfs_board_eval_fn board_eval;
    void* move_user;
    bfs_move_start_fn move_start;    // both null: moves are scored np.linspace(0, 1, total) over the
    bfs_move_finish_fn move_finish;  // batch, the reference's fake policy (code_testing.py:634-640)
    size_t max_nodes;           // arena limit in nodes (0.8 KB each), checked before the arena grows; 0 = 2^22 (3.5 GB);
                                // at most INT_MAX (node ids are ints): more is refused by bfs_search_create
    int threads;                // host threads for the per-node loops of a batch (the reference's prange,
                                // numba_board.py:242-266); 0 = the job's cores (OMP_NUM_THREADS if set, else the
                                // machine's) minus four for the caller, the evaluator lanes and the GPU runtime, at
                                // most 16; results do not depend on it
} bfs_config;

typedef struct {
    uint64_t iterations;        // do_iteration calls
    uint64_t nodes_out;         // nodes taken from the open list and expanded ("total_out")
    uint64_t nodes_in;          // nodes inserted into the open list ("total_in")
    uint64_t boards_evaluated;  // boards sent to the value network
    uint64_t moves_scored;      // moves sent to the move-scoring network
    uint64_t max_batch;         // largest evaluation batch
    double seconds;             // wall time inside search calls
    double eval_wait_seconds;   // of which waiting for the evaluators
    uint64_t eval_calls;        // value-network calls (boards_evaluated / eval_calls = mean evaluation batch)
    uint64_t policy_calls;      // move-scoring calls
    // where the wall time of do_iteration goes (seconds; their sum + select_seconds ~ seconds):
    double expand_seconds;      // children made, depth-0 children checked and packed into records (threads)
    double settle_seconds;      // other children: draw / table / mate checks, move generation (threads)
    double policy_seconds;      // move-scoring: packing (threads) + both evaluator calls
    double value_wait_seconds;  // waiting for the value network after all of the above
    double update_seconds;      // scores into the tree and the transposition table (one thread, beside the next two
                                // stages): what was left of it when they were done
    double compact_seconds;     // open children kept, re-insertion lists
    double select_seconds;      // open list: insert_and_get_batch
    uint64_t arena_peak;        // most nodes the arena held at once
    double update_busy_seconds; // the tree update's own time (overlaps policy_seconds and compact_seconds)
    // inside select_seconds: binning + insertion, taking candidates off the bins, the candidates' validity checks
    double select_insert_seconds, select_pop_seconds, select_live_seconds;
    // inside compact_seconds: the copy of the open children to the arena
    double compact_copy_seconds;
    uint64_t wide_iterations;   // batches of at least `wide_batch` nodes: their loops ran on every pool thread
    uint64_t overlapped_updates;// of which the tree update ran on the evaluation lane beside the hand-over of the
                                // move scores and the arena copy (four host threads or more)
} bfs_stats;

typedef struct bfs_search bfs_search;

const char* bfs_last_error(void);

// board primitives (for the perft / legality tests, code_testing.py:64-83, :266-293)
int bfs_perft(const char* fen, int depth, uint64_t* nodes);
int bfs_legal_moves(const char* fen, uint8_t* moves /* [100][3] */, int* n_moves);
int bfs_legal_moves_record(const void* record82, uint8_t* moves /* [100][3] */, int* n_moves);
int bfs_push(const char* fen, const uint8_t* move /* [3] */, char* fen_out, size_t cap);
// is_legal_move on arbitrary (from, to, promotion) triples, decided two ways: *by_generator = the move is
// among the generated legal moves (what the search uses to vet a table move), *direct = decided from the
// move alone (numba_board.py:1149-1151).  The reference's sweep: code_testing.py:55-59, :266-293.
int bfs_is_legal(const char* fen, const uint8_t* moves /* [n][3] */, size_t n, uint8_t* by_generator, uint8_t* direct);
// child record of a node record after a legal move (push_moves on one record, numba_board.py:242-417)
int bfs_push_record(const void* record82, const uint8_t* move /* [3] */, void* record88_out);
// zobrist_hash_test, code_testing.py:86-136: random legal lines of at most `plies` moves from `fen`; at every
// ply the incrementally maintained hash is compared with the hash computed from scratch.
int bfs_hash_walk(const char* fen, int lines, int plies, uint64_t seed, uint64_t* positions_checked, uint64_t* mismatches);
int bfs_record_from_fen(const char* fen, void* record88);
int bfs_random_playout_records(const char* fen, int plies, uint64_t seed, void* records88, int* n_records);

// reference fakes and oracle (code_testing.py:516-545, :548-607), natively
int bfs_material_eval(void* user, const void* records, size_t stride, const uint8_t* mask, size_t n, float* scores,
                      size_t* n_scored);
int bfs_negamax(const char* fen, int depth, bfs_board_eval_fn eval, void* user, float* value);

int bfs_search_create(bfs_search** out, const bfs_config* cfg);
void bfs_search_destroy(bfs_search* s);
int bfs_search_threads(const bfs_search* s);   // host threads the instance runs its per-node loops on
void bfs_search_clear_tt(bfs_search* s);
void bfs_search_stats(const bfs_search* s, bfs_stats* out, int reset);
// zero_window_search of code_testing.py:598-625: value of the position searched to `depth` against `separator`
int bfs_zero_window(bfs_search* s, const char* fen, int depth, float separator, float* value);
int bfs_mtd_f(bfs_search* s, const char* fen, int depth, float first_guess, float guess_increment, float* value,
              uint8_t* move /* [3] */);
int bfs_iterative_deepening(bfs_search* s, const char* fen, const int* depths, int n_depths, float* value,
                            uint8_t* move /* [3] */);
}
