// Host-sanitizer driver for the threaded search (tests/test_search_sanitizers.py): two iterative-deepening searches on
// four host threads with the material-count evaluator; built with -fsanitize=thread and -fsanitize=address,undefined.
#include <cstdio>
#include <vector>
#include "search.h"
int main() {
    std::vector<float> edges(1000);
    for (int i = 0; i < 1000; ++i) edges[i] = -4000.f + 8000.f * i / 999.f;
    bfs_config cfg{};
    cfg.max_batch_size = 256; cfg.tt_bits = 16; cfg.bin_edges = edges.data(); cfg.n_bin_edges = edges.size();
    cfg.board_eval = bfs_material_eval; cfg.threads = 4;
    const char* fens[] = {"rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
                          "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"};
    for (const char* fen : fens) {
        bfs_search* s = nullptr;
        if (bfs_search_create(&s, &cfg)) { printf("create failed: %s\n", bfs_last_error()); return 1; }
        int depths[] = {1, 2, 3, 4};
        float v; uint8_t mv[3];
        if (bfs_iterative_deepening(s, fen, depths, 4, &v, mv)) { printf("search failed: %s\n", bfs_last_error()); return 1; }
        bfs_stats st; bfs_search_stats(s, &st, 0);
        printf("value %.1f move %d %d nodes %llu iterations %llu\n", v, mv[0], mv[1], (unsigned long long)st.nodes_out, (unsigned long long)st.iterations);
        bfs_search_destroy(s);
    }
    return 0;
}
