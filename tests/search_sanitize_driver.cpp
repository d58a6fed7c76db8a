// Host-sanitizer driver for the threaded search (tests/test_search_sanitizers.py): two iterative-deepening searches on
// four host threads with the material-count evaluator; built with -fsanitize=thread and -fsanitize=address,undefined.
// The batch limit is large enough for batches of 320 nodes and more, which take the search's "wide" path: every
// pool thread (the second worker group included) and the tree update on the evaluation lane beside score_end and
// the arena copy.  The counts of such batches are printed so that the test can insist that the path ran.
#include <cstdio>
#include <vector>
#include "search.h"
// a stand-in for the move-scoring network, so that the policy lane runs too: scores from the move's squares
static int fake_start(void*, const void*, size_t, const uint8_t*, size_t) { return 0; }
static int fake_finish(void*, const uint8_t* from, const uint8_t* filt, const uint8_t*, size_t, size_t n_moves, float* scores) {
    for (size_t i = 0; i < n_moves; ++i) scores[i] = (float)((from[i] * 73 + filt[i]) % 97) / 97.0f;
    return 0;
}

int main() {
    std::vector<float> edges(1000);
    for (int i = 0; i < 1000; ++i) edges[i] = -4000.f + 8000.f * i / 999.f;
    bfs_config cfg{};
    cfg.max_batch_size = 2048; cfg.tt_bits = 16; cfg.bin_edges = edges.data(); cfg.n_bin_edges = edges.size();
    cfg.board_eval = bfs_material_eval; cfg.threads = 4;
    cfg.move_start = fake_start; cfg.move_finish = fake_finish;
    const char* fens[] = {"rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
                          "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"};
    for (const char* fen : fens) {
        bfs_search* s = nullptr;
        if (bfs_search_create(&s, &cfg)) { printf("create failed: %s\n", bfs_last_error()); return 1; }
        int depths[] = {1, 2, 3, 4};
        float v; uint8_t mv[3];
        if (bfs_iterative_deepening(s, fen, depths, 4, &v, mv)) { printf("search failed: %s\n", bfs_last_error()); return 1; }
        bfs_stats st; bfs_search_stats(s, &st, 0);
        printf("value %.1f move %d %d nodes %llu iterations %llu wide %llu overlapped %llu\n", v, mv[0], mv[1],
               (unsigned long long)st.nodes_out, (unsigned long long)st.iterations, (unsigned long long)st.wide_iterations,
               (unsigned long long)st.overlapped_updates);
        bfs_search_destroy(s);
    }
    return 0;
}
