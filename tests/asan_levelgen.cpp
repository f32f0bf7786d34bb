// Host build of the level generators under ASan/UBSan (levelgen_core.h is the code k_levelgen runs on the GPU).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mgx.h"
#include "mgx_internal.h"
static char g_err[512];
int mgx_fail(int status, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return status; }
int main(int argc, char **argv)
{
    const int n_seeds = argc > 1 ? atoi(argv[1]) : 200;
    long total = 0;
    for (int i = 0;; i++) {
        const char *id = mgx_env_id(i);
        if (!id) break;
        mgx_config cfg;
        if (mgx_env_config(id, &cfg)) { printf("config failed: %s\n", id); return 1; }
        const size_t cells = (size_t)cfg.width * cfg.height;
        std::vector<uint64_t> seeds(n_seeds);
        for (int k = 0; k < n_seeds; k++) seeds[k] = (uint64_t)k * 0x9E3779B97F4A7C15ull + (uint64_t)i * 1000003ull + (k < 20 ? 0 : (1ull << 40));
        for (int k = 0; k < 20; k++) seeds[k] = (uint64_t)k;
        std::vector<uint8_t> grid(n_seeds * cells * 3);
        std::vector<int32_t> agent(n_seeds * 3);
        std::vector<uint32_t> task(n_seeds);
        int rc = mgx_generate_levels_ex(&cfg, n_seeds, seeds.data(), grid.data(), agent.data(), task.data());
        if (rc) { printf("%s: generate_levels rc=%d %s\n", id, rc, g_err); return 1; }
        const int K = 12;
        std::vector<uint8_t> sg(K * cells * 3);
        std::vector<int32_t> sa(K * 3);
        std::vector<uint32_t> st(K);
        for (int k = 0; k < 8; k++) {
            rc = mgx_generate_level_stream_ex(&cfg, seeds[k * 7 % n_seeds], K, sg.data(), sa.data(), st.data());
            if (rc) { printf("%s: level_stream rc=%d %s\n", id, rc, g_err); return 1; }
        }
        total += n_seeds + 8 * K;
    }
    printf("asan levelgen ok: %ld levels\n", total);
    return 0;
}
