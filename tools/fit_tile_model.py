"""Fit pick_tile()'s cost model (round cost = a + b*K microseconds per tile config) to a `gemm_lab model` sweep and report the
regret of argmin(model) against the per-shape best config.  python tools/fit_tile_model.py gpurun_out/model.jsonl"""
import json
import sys
import numpy as np

TILES = [(128, 128, 512), (128, 80, 256), (64, 128, 256), (64, 64, 256), (256, 256, 256), (256, 320, 256), (256, 160, 256),
         (256, 256, 256), (256, 320, 256)]  # (bm, bn, slots); 7 / 8 = ping-pong 256x256 / 256x320


def rounds(M, N, c):
    bm, bn, slots = TILES[c]
    tiles = -(-M // bm) * -(-N // bn)
    return -(-tiles // slots)


def main():
    recs = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
    fit = {}
    for kind in ("linear", "conv"):
        rs = [r for r in recs if r["kind"] == kind]
        for c in range(len(TILES)):
            rows, y = [], []
            for r in rs:
                n = rounds(r["M"], r["N"], c)
                t = r["us"][c] / n
                w = 1.0 / t  # relative error
                rows.append([w, w * r["K"]])
                y.append(w * t)
            sol, *_ = np.linalg.lstsq(np.array(rows), np.array(y), rcond=None)
            fit[(kind, c)] = (max(sol[0], 0.1), sol[1])
    for c in range(len(TILES)):
        print("cfg %d  %dx%d: lin a=%.2f b=%.5f | conv a=%.2f b=%.5f" % (c, TILES[c][0], TILES[c][1], *fit[("linear", c)], *fit[("conv", c)]))
    print("    {" + "},\n    {".join("%d, %d, %s, %d, %.2ff, %.5ff, %.2ff, %.5ff" % (TILES[c][0], TILES[c][1], "true" if TILES[c][1] % 32 == 0 and TILES[c][1] // (4 if TILES[c][0] == 256 and TILES[c][1] != 160 else 2) % 32 == 0 else "false", TILES[c][2], *fit[("linear", c)], *fit[("conv", c)]) for c in range(len(TILES))) + "}")
    tot_best = tot_pick = 0.0
    worst = (0, None)
    for r in recs:
        pred = [rounds(r["M"], r["N"], c) * (fit[(r["kind"], c)][0] + fit[(r["kind"], c)][1] * r["K"]) for c in range(len(TILES))]
        pick = int(np.argmin(pred))
        best = int(np.argmin(r["us"]))
        tot_best += r["us"][best]
        tot_pick += r["us"][pick]
        reg = r["us"][pick] / r["us"][best] - 1
        if reg > worst[0]:
            worst = (reg, r, pick, best)
        if reg > 0.03:
            print("  regret %4.1f%% %s M%d N%d K%d pick %d (%.1f us) best %d (%.1f us)" % (100 * reg, r["kind"], r["M"], r["N"], r["K"], pick, r["us"][pick], best, r["us"][best]))
    print("sum of picks / sum of bests = %.4f" % (tot_pick / tot_best))


if __name__ == "__main__":
    main()
