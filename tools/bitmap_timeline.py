"""Where a bitmap_kernel workgroup's time goes (round 5, profiles/r05_bitmap.txt).
    tools/build_variant.sh bmtl -DF110_BM_TIMELINE            # VARIANT_DIR=variants_ship to take it to the GPU box
    F110_LIB=build_variants/bmtl.so F110_LIB_OLDER=1 F110_BM_TL_DUMP=/tmp/tl.bin python tools/bench_bitmap.py --reps 1
    python tools/bitmap_timeline.py /tmp/tl.bin [workgroups of the launch, default 768]
The diagnostics build stamps the 100 MHz clock at every stage boundary of every image ([n][10] uint64) and prints the mean
stage times itself; this script reads the dump: the lives of the launch's workgroups (workgroup g draws images g, g + grid,
...), their spread, and whether the slow ones are slow because of their images or because of where they sit."""
import numpy as np, sys
h = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 10).astype(np.int64)
n = len(h); grid = int(sys.argv[2]) if len(sys.argv) > 2 else 768
t0 = h[:, 0].min()
life = []; busy = []
for g in range(grid):
    idx = np.arange(g, n, grid)
    s = h[idx]
    life.append((s[-1, 7] - s[0, 0]) / 100.0)
    busy.append(((s[:, 7] - s[:, 0]).sum()) / 100.0)
life = np.array(life); busy = np.array(busy)
print('life mean %.1f min %.1f max %.1f; busy (sum of image times) mean %.1f; corr(life, busy) %.3f' % (life.mean(), life.min(), life.max(), busy.mean(), np.corrcoef(life, busy)[0, 1]))
for m in (8, 32, 256):
    grp = [life[np.arange(grid) % m == k].mean() for k in range(m)]
    print('by g %% %d: min %.1f max %.1f spread of group means' % (m, min(grp), max(grp)))
grp8 = [round(life[np.arange(grid) % 8 == k].mean(), 1) for k in range(8)]
print('g % 8 means:', grp8)
# per-image duration by position in the launch
dur = (h[:, 7] - h[:, 0]) / 100.0
print('image time mean %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f' % (dur.mean(), *np.percentile(dur, [10, 50, 90, 99]), dur.max()))
# slow workgroups: are their images slow throughout?
slow = np.argsort(life)[-20:]; fast = np.argsort(life)[:20]
print('slowest 20 workgroups: mean image time %.2f; fastest 20: %.2f' % (np.mean([dur[np.arange(g, n, grid)].mean() for g in slow]), np.mean([dur[np.arange(g, n, grid)].mean() for g in fast])))
print('slowest ids', sorted(slow.tolist())[:20]); print('fastest ids', sorted(fast.tolist())[:20])
