// How many MT19937 words does `seed(s); reset()` of each built-in id consume?  (Sizing of the seed kernels' first-words window,
// csrc/k_levelgen.hip.)  g++ -std=c++17 -O2 -D__host__= -D__device__= -Iinclude -Igym-minigrid_amd/csrc tools/draw_stats.cpp -o /tmp/draw_stats
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../gym-minigrid_amd/csrc/levelgen.cpp"
static char g_err[512];
int mgx_fail(int status, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return status; }
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    for (int i = 0;; i++) {
        const char *id = mgx_env_id(i);
        if (!id) break;
        mgx_config cfg;
        mgx_env_config(id, &cfg);
        if (!lg_uses_rng(cfg)) { printf("%-44s no draws\n", id); continue; }
        std::vector<int> d(n);
        Rng rng;
        for (int k = 0; k < n; k++) {
            LgLevel L;
            int16_t ws[MGX_LG_WS_WORDS];
            LgCmd cmds[MGX_LG_MAX_CMDS];
            L.cmds = cmds; L.ncmd = 0; L.W = cfg.width; L.H = cfg.height; L.ax = L.ay = -1; L.adir = 0; L.ws = ws;
            rng.seed_gym((uint64_t)k * 2654435761ull + 12345);
            rng.next32(); rng.idx = 0; // first twist done, position 0
            int blocks = 0, last = 0;
            // count words: idx wraps to 0 on each new block
            Rng *r = &rng;
            struct Counting { Rng *r; long n; bool alive() const { return true; } uint32_t next32() { n++; return r->next32(); }
                uint32_t bounded(uint32_t m) { if (!m) return 0; uint32_t mask = m; mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; uint32_t v; do { v = next32() & mask; } while (v > m); return v; }
                int randint(int lo, int hi) { return lo + (int)bounded((uint32_t)(hi - lo - 1)); } } c{r, 0};
            (void)blocks; (void)last;
            lg_generate(cfg, c, L);
            d[k] = (int)c.n;
        }
        std::sort(d.begin(), d.end());
        double mean = 0; for (int v : d) mean += v; mean /= n;
        printf("%-44s mean %7.1f  p50 %5d  p99 %5d  p99.9 %5d  max %6d   >32: %6.3f%%  >64: %6.3f%%  >96: %6.3f%%\n", id, mean, d[n / 2], d[(int)(n * 0.99)], d[(int)(n * 0.999)], d[n - 1],
               100.0 * (d.end() - std::upper_bound(d.begin(), d.end(), 32)) / n, 100.0 * (d.end() - std::upper_bound(d.begin(), d.end(), 64)) / n,
               100.0 * (d.end() - std::upper_bound(d.begin(), d.end(), 96)) / n);
    }
    return 0;
}
