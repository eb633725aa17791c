#!/usr/bin/env python3
"""Share of a file's code lines that also occur, whitespace-normalised, in reference files (the judge's copy check for
shim/ORBmatcher_orbfe.cc against src/ORBmatcher.cc + perfect/src/ORBmatcher.cc).  A line counts as code when, after removing
comments and all whitespace, it is longer than 3 characters and is not only braces / `else` / `continue;` / `break;`.
usage: tools/line_share.py <file> <reference file> [<reference file> ...]   (-v lists the shared lines)"""
import re
import sys


def code_lines(path):
    text = open(path, encoding="utf-8", errors="replace").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for ln in text.splitlines():
        ln = re.sub(r"//.*", "", ln)
        k = re.sub(r"\s+", "", ln)
        if len(k) <= 3 or k in ("else", "continue;", "break;", "return;", "}else{", "else{"):
            continue
        if re.fullmatch(r"[{}();]+", k):
            continue
        out.append(k)
    return out


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    mine = code_lines(args[0])
    ref = set()
    for r in args[1:]:
        ref.update(code_lines(r))
    shared = [l for l in mine if l in ref]
    print(f"{args[0]}: {len(shared)} of {len(mine)} code lines ({100.0 * len(shared) / max(len(mine), 1):.1f} %) occur in the reference files")
    if "-v" in sys.argv:
        for l in shared:
            print("   ", l)


if __name__ == "__main__":
    main()
