"""Report on a k_step wave timeline dump (tuning aid; the instrumented build is tools/build_variant.sh tl -DMGX_TIMELINE=1,
the dump comes from MGX_TL_FILE=... MGX_TL_LAUNCH=n under any driver).  Per wave: s_memrealtime (100 MHz) at entry, tile staged,
transition done, observation computed, stores issued; HW_ID; sub-phase offsets."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
t = a[:, :5].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) * 0.01  # us
hw = a[:, 5]
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 32) & 15) << 7)  # cu_id | se_id | xcc
sub = np.stack([(a[:, 6] >> s) & 0xFFFFF for s in (0, 20)], axis=1).astype(np.int64) * 0.01  # since "tile staged": forward-cell index known / transition applied
osub = np.stack([(a[:, 7] >> s) & 0xFFFFF for s in (0, 20, 40)], axis=1).astype(np.int64) * 0.01  # since "transition done": view gathered / occlusion applied / triples decoded
simd = (hw >> 4) & 3
print("waves %d   kernel (first entry -> last store issued) %.2f us" % (len(t), t[:, 4].max()))
names = ["stage tile (loads -> LDS)", "transition", "observation compute", "store issue"]
q = [5, 50, 95]
print("%-28s %s" % ("entry time (us)", np.percentile(t[:, 0], [0, 5, 50, 95, 100]).round(2)))
first = t[:, 0] < 3.0
for lab, m in (("waves entering < 3 us", first), ("waves entering later", ~first)):
    if not m.any():
        continue
    print("%s: %d" % (lab, m.sum()))
    for i, n in enumerate(names):
        d = t[m, i + 1] - t[m, i]
        print("   %-28s mean %6.2f   p5/p50/p95 %s" % (n, d.mean(), np.percentile(d, q).round(2)))
    tr = t[m, 2] - t[m, 1]
    for lab2, lo, hi in (("  . unpack + bounds", None, 0), ("  . forward cell + action switch", 0, 1), ("  . stores, counters, reset, record", 1, None)):
        d = (sub[m, hi] if hi is not None else tr) - (sub[m, lo] if lo is not None else 0)
        print("   %-34s mean %6.2f   p5/p50/p95 %s" % (lab2, d.mean(), np.percentile(d, q).round(2)))
    oc = t[m, 3] - t[m, 2]
    for lab2, lo, hi in (("  . bounds + 49-cell gather", None, 0), ("  . occlusion", 0, 1), ("  . decode to triples", 1, 2), ("  . byte phase + LDS image", 2, None)):
        d = (osub[m, hi] if hi is not None else oc) - (osub[m, lo] if lo is not None else 0)
        print("   %-34s mean %6.2f   p5/p50/p95 %s" % (lab2, d.mean(), np.percentile(d, q).round(2)))
    life = t[m, 4] - t[m, 0]
    print("   %-28s mean %6.2f   p5/p50/p95 %s" % ("lifetime", life.mean(), np.percentile(life, q).round(2)))
    print("   %-28s p5/p50/p95/max %s" % ("end time", np.percentile(t[m, 4], [5, 50, 95, 100]).round(2)))
# occupancy of each phase over time
T = t[:, 4].max()
edges = np.arange(0, T + 1.0, 1.0)
print("\nwaves in each phase at the start of every microsecond (stage / transition / obs compute / store issue):")
for e in edges:
    row = [int(((t[:, i] <= e) & (e < t[:, i + 1])).sum()) for i in range(4)]
    print("  t=%5.1f  %6d %6d %6d %6d   total %6d" % (e, row[0], row[1], row[2], row[3], sum(row)))
# per-SIMD view of the first round: how spread are the compute-done times of the waves sharing one SIMD
key = cu * 4 + simd
spread = []
for k in np.unique(key):
    m = (key == k) & first
    if m.sum() >= 4:
        spread.append(t[m, 3].max() - t[m, 3].min())
if spread:
    print("\nfirst-round waves sharing a SIMD: spread of their 'observation computed' times: mean %.2f us (p5 %.2f, p95 %.2f), %d SIMDs" %
          (np.mean(spread), np.percentile(spread, 5), np.percentile(spread, 95), len(spread)))
