"""Counts the instructions one steady-state row step of k_fast_map / k_blur7 issues (VALU ceiling of bench.py's roofline).
usage: python tools/valu_count.py      (needs hipcc; compiles csrc/orbfe_kernels.hip to assembly for gfx950)

k_fast_map's row loop is unrolled 8-fold; the instruction mix of the whole unrolled body (from the loop header to the
back edge, emission slow paths included) divided by 8 is reported.  The numbers are pasted into bench.py (VALU_MODEL)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_slam2_ssd_semantic_amd", "csrc")


def assembly():
    out = os.path.join(tempfile.gettempdir(), "orbfe_kernels.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950",
                           "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                           os.path.join(CSRC, "orbfe_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def function_body(lines, prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or
                 (l.startswith(prefix) and ": " in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def mix(body):
    c = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "other": 0}
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["other"] += 1
    return c


def main():
    lines = assembly()
    for name, unroll in (("_Z10k_fast_mapILi0E", 8), ("_Z10k_fast_mapILi1E", 8)):
        body = function_body(lines, name)
        # the row loop = the outermost loop: from the first loop header to the last backward branch to it
        hdr = [i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)]
        back = {}
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                tgt = next((j for j in hdr if body[j].startswith(m.group(1) + ":")), None)
                if tgt is not None and tgt < i:
                    back[tgt] = max(back.get(tgt, 0), i)
        lo, hi = max(back.items(), key=lambda kv: kv[1] - kv[0])
        c = mix(body[lo:hi + 1])
        print(name, "row loop lines", lo, hi, {k: round(v / unroll, 1) for k, v in c.items()}, "per row step;",
              "whole kernel", mix(body))


if __name__ == "__main__":
    sys.exit(main())
