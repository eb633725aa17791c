"""One CPU-baseline worker process (TEST / BENCH INFRASTRUCTURE): oracle extract + BF match to the previous frame on its
own stream of S(seed) frames, from a common wall-clock tick for a fixed duration; prints the number of frames done.
Started by bench.py's cpu_baseline_all_cores leg, one per host core.  usage: cpu_worker.py w h nfeat t_go duration idx"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")

from oracle import oracle_ffi as O  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frame  # noqa: E402


def main():
    w, h, nf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    t_go, dur, idx = float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
    e = None
    if os.environ.get("ORBFE_CPU_WORKER_KIND") == "reference":   # the unmodified reference ORBextractor.cc (oracle/_ref)
        try:
            from oracle import ref_ffi as R
            R.configure(bump=True, canonical_trig=True, blur_mode=0)
            e = R.RefExtractor(nf, 1.2, 8, 20, 7)
        except Exception:
            e = None
    if e is None:
        e = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    frames = [synth_frame(20000 + 4 * idx + i, h, w) for i in range(2)]
    prev = e(frames[0])  # warm-up
    late = time.time() >= t_go   # this interpreter came up after the common tick: its window is shorter, say so
    while time.time() < t_go:
        pass
    n = 0
    t_end = t_go + dur
    while time.time() < t_end:
        k, d = e(frames[n & 1])
        O.match_bf(d, prev[1], k["angle"], prev[0]["angle"], 0.9, 100, True)
        prev = (k, d)
        n += 1
    print(("late " if late else "") + str(n))


if __name__ == "__main__":
    main()
