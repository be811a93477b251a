#!/usr/bin/env python3
"""Instruction census of `k_accumulate` from the gfx950 ISA the build produces (hipcc -S of csrc/msm_pippenger.hip with
the build's flags): the issue roofline of an issue-bound kernel is instructions per entry, so this is the number to
drive (1 475 multiply-adds in 2 276 issued instructions = 65 % useful slots).

Counts the static instructions of the accumulate loop (two entries per trip: loop total / 2 = per entry) by class and
of the two inlined mixed-addition blocks, and writes profiles/<tag>_isa_k_accumulate.json with the kernel-source hash;
bench.py quotes it as `issue_roofline` only while the hash matches the tree.
Usage: python tools/isa_stats.py [--tag r02] [--kernel k_accumulate] [--keep-asm path]"""
import argparse
import importlib.util
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "snark-verifier_amd")


def _mod(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


CLASSES = [
    ("mad64", re.compile(r"^v_mad_(i64_i32|u64_u32)\b")),
    ("mul_lo", re.compile(r"^v_mul_(lo_u32|hi_u32|lo_i32|hi_i32|u32_u24|i32_i24)\b")),
    ("s_nop", re.compile(r"^s_nop\b")),
    ("mov", re.compile(r"^v_(mov_b32|mov_b64|accvgpr_\w+|cndmask_b32|readfirstlane_b32|readlane_b32|writelane_b32)\b")),
    ("mask_and", re.compile(r"^v_(and_b32|and_or_b32|bfe_[iu]32|bfi_b32)\b")),
    ("shift", re.compile(r"^v_(ashrrev_i64|lshrrev_b64|lshlrev_b64|alignbit_b32|ashrrev_i32|lshrrev_b32|lshlrev_b32|lshl_add_u64|lshl_add_u32|lshl_or_b32)\b")),
    ("addsub", re.compile(r"^v_(add|sub|subrev|addc|subb|subbrev)(_co)?(_ci)?_[iu](32|16)\b|^v_(add3_u32|add_lshl_u32|sub_nc_u32)\b")),
    ("cmp", re.compile(r"^v_cmp")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")),
    ("lds", re.compile(r"^ds_")),
    ("waitcnt", re.compile(r"^s_waitcnt")),
    ("branch", re.compile(r"^s_(cbranch|branch)")),
    ("salu", re.compile(r"^s_")),
]


def classify(mn):
    mn = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    for name, rx in CLASSES:
        if rx.match(mn):
            return name
    return "other_valu" if mn.startswith("v_") else "other"


def census(lines):
    out = {}
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        mn = s.split()[0]
        if mn.endswith(":"):
            continue
        c = classify(mn)
        out[c] = out.get(c, 0) + 1
        out["total"] = out.get("total", 0) + 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--kernel", default="k_accumulate")
    ap.add_argument("--keep-asm")
    ap.add_argument("--out")
    args = ap.parse_args()
    build = _mod("_b", os.path.join(PKG, "build.py"))
    srchash = _mod("_h", os.path.join(PKG, "_srchash.py")).kernel_source_hash()
    asm = args.keep_asm or os.path.join(tempfile.mkdtemp(), "msm_pippenger.s")
    cmd = [build.HIPCC] + build.FLAGS + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "msm_pippenger.hip"), "-o", asm]
    subprocess.run(cmd, check=True, capture_output=True)
    text = open(asm).read().split("\n")
    # the kernel itself (a template instantiation `...k_accumulateILi64ELb0EE...` or a plain `...k_fooE...`), not a longer name
    # that merely starts with it (k_accumulate_pairs); of a template the <.., false> (packed point table) instantiation first
    cands = [i for i, l in enumerate(text) if re.match(r"^_ZN6snarkv\d+%s[IE]\w*:" % args.kernel, l)]
    start = next((i for i in cands if "Lb0" in text[i]), cands[0])
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start + 1:end]
    # the loop: every basic block annotated "in Loop: Header=BBn_m" or "Inner Loop Header"
    hdr = next((re.search(r"(\.LBB\d+_\d+):.*Inner Loop Header", l) for l in body if "Inner Loop Header" in l), None)
    if not hdr:
        raise SystemExit("no loop found in " + args.kernel)
    label = hdr.group(1)[2:]  # BBn_m
    in_loop, loop_lines, cur = False, [], False
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = ("Header=" + label) in m.group(2) or (m.group(1)[2:] == label)
        if cur:
            loop_lines.append(l)
    loop = census(loop_lines)
    # the straight-line blocks of the loop above 1 500 instructions: the inlined mixed additions
    blocks, acc = [], []
    for l in loop_lines + [".LBB_end:"]:
        if re.match(r"^\.LBB", l):
            c = census(acc)
            if c.get("total", 0) >= 1500:
                blocks.append(c)
            acc = []
        else:
            acc.append(l)
    entries_per_trip = 2
    per_entry = {k: v / entries_per_trip for k, v in loop.items()}
    rec = {
        "kernel": args.kernel, "kernel_source_hash": srchash, "flags": build.FLAGS,
        "entries_per_loop_trip": entries_per_trip,
        "loop_static_instructions": loop, "per_entry": per_entry,
        "madd_blocks": blocks,
        "mads_per_entry": per_entry.get("mad64", 0.0), "instr_per_entry": per_entry.get("total", 0.0),
        "useful_issue_fraction": per_entry.get("mad64", 0.0) / max(1.0, per_entry.get("total", 0.0)),
        "whole_kernel_static": census(body),
        "note": "static census of the accumulate loop (both entries of a trip; rare flush paths included), from "
                "hipcc -S with the build's flags; on gfx950 every VALU instruction of this mix costs one issue slot "
                "(profiles/r01_ubench_isa_rates.txt), so useful_issue_fraction = multiply-adds / instructions",
    }
    out = args.out or os.path.join(ROOT, "profiles", "%s_isa_%s.json" % (args.tag, args.kernel))
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("kernel_source_hash", "mads_per_entry", "instr_per_entry", "useful_issue_fraction")}))
    print("per entry:", {k: round(v, 1) for k, v in sorted(per_entry.items(), key=lambda kv: -kv[1])})
    print("madd blocks:", [b.get("total") for b in blocks], "->", out)


if __name__ == "__main__":
    main()
