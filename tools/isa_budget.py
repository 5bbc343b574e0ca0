#!/usr/bin/env python
"""Per-(row, splat)-step instruction budget of the compositing kernels, read off the ISA the product build ships (needs no GPU).

    python tools/isa_budget.py [--steps fwd=264000,bwd=264000]

For every kernel of composite.hip whose main loop evaluates a splat per step (marker: the v_exp_f32 of the Gaussian's power), the
instructions between two consecutive markers of the same loop -- one step in steady state: the tail of step k and the head of step k + 1 of
the pair-unrolled loop -- are classified by issue port and priced with the issue intervals measured on MI355X by tools/ubench/valu_rate.hip
(profiles/r04_valu_rate_ubench.txt; per SIMD, one wave: plain VALU 1.34 ns, DPP 1.91, transcendental 3.6, v_mad_u64_u32 2.16; SALU / LDS /
VMEM issue from their own ports and overlap with other waves' VALU).  floor_us = wave steps x VALU ns per step / 1024 SIMDs: the time the
launch would take if the vector pipes never idled and the work were spread evenly -- the instruction-count lever's limit."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mm3dgs_slam_amd", "csrc", "composite.hip")
NS = {"valu": 1.34, "dpp": 1.91, "trans": 3.6, "mad64": 2.16}


def disassemble():
    co = "/tmp/isa_budget.co"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                           "-fno-slp-vectorize", "-c", SRC, "-o", co], stderr=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={co}",
                           f"--output={co}.elf", "--unbundle"])
    return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f"{co}.elf"], text=True)


def classify(op, text):
    if op.startswith("v_"):
        if "dpp" in op or "quad_perm" in text or "row_" in text:
            return "dpp"
        if re.match(r"v_(exp|rcp|rsq|sqrt|log|sin|cos)_", op):
            return "trans"
        if op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
            return "mad64"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    steps = {"fwd": 264000, "bwd": 264000}
    for a in sys.argv[1:]:
        if a.startswith("--steps"):
            for kv in sys.argv[sys.argv.index(a) + 1].split(","):
                k, v = kv.split("=")
                steps[k] = int(v)
    kernels, cur = collections.OrderedDict(), None
    for line in disassemble().splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\w+)\s*(.*?)\s*//", line)
        if m and cur is not None:
            cur.append((m.group(1), m.group(2)))
    name_of = {"sort_composite_fwd_kernelILi6E": ("sort + forward compositor", "fwd"), "composite_bwd_kernelILi6ELi1E": ("mapping backward compositor", "bwd"),
               "composite_bwd_kernelILi6ELi2E": ("tracking backward compositor", "bwd"), "sort_composite_fwd_bwd_track": ("fused tracking kernel", None),
               "composite_bwd_kernelILi3ELi0E": ("generic backward compositor, C = 3", None), "composite_fwd_kernelILi3E": ("generic forward compositor, C = 3", None)}
    print(f"{'kernel':44s} {'loop':>11s} {'VALU':>5s} {'DPP':>4s} {'trans':>5s} {'mad64':>5s} {'SALU':>5s} {'LDS':>4s} {'VMEM':>5s} {'VALU ns/step':>13s} {'floor us':>9s}")
    for mangled, ins in kernels.items():
        label = next((v for k, v in name_of.items() if k in mangled), None)
        if label is None:
            continue
        marks = [i for i, (op, _) in enumerate(ins) if op.startswith("v_exp_f32")]
        loop, seen = 0, set()
        for a, b in zip(marks, marks[1:]):
            if b - a > 125:
                continue          # markers of different loops (a step is 50 - 100 instructions)
            c = collections.Counter(classify(op, text) for op, text in ins[a:b])
            ns = sum(c[k] * NS[k] for k in NS)
            is_bwd = c["dpp"] > 0
            # round 6: the backward loops are two-phase -- the loop found here (no DPP in it) is PHASE 1 of a backward kernel (lane = pixel: alpha, T, dL/dalpha ->
            # (u, w) into the wave's LDS tile); phase 2 (lane = (entry, part of the block), once per 8 / 4 entries) is priced in DESIGN.md section 4
            p1 = (not is_bwd) and (label[1] == "bwd" or "backward" in label[0] or ("tracking kernel" in label[0] and "fwd" in seen))
            if p1:
                tag = "bwd ph.1 " + ("gen." if "p1" not in seen else "fast")
                if "p1b" in seen:
                    continue
                seen.add("p1b" if "p1" in seen else "p1")
                print(f"{label[0]:44s} {tag:>11s} {c['valu']:5d} {c['dpp']:4d} {c['trans']:5d} {c['mad64']:5d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:5d} "
                      f"{ns:13.1f} {steps['bwd'] * ns / 1024 / 1e3:9.1f}")
                continue
            if not is_bwd and "fwd" in seen:
                continue
            seen.add("fwd" if not is_bwd else "bwd")
            n_steps = steps["bwd" if is_bwd else "fwd"]
            loop += 1 if is_bwd else 0
            inst = ("general" if loop == 1 else "fast") if is_bwd else "-"      # backward: the general loop instance, then the one for waves whose pixels carry no silhouette / depth^2 / background gradient (the SLAM losses)
            print(f"{label[0]:44s} {('bwd ' + inst if is_bwd else 'fwd'):>11s} {c['valu']:5d} {c['dpp']:4d} {c['trans']:5d} {c['mad64']:5d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:5d} "
                  f"{ns:13.1f} {n_steps * ns / 1024 / 1e3:9.1f}")
    print(f"(wave steps per launch: forward {steps['fwd']}, backward {steps['bwd']} -- tools/xcd_balance.py on the benchmark map at frame 8: 263 654 wave steps for 992 024 row steps;\n"
          " lanes useful per (row, splat) step: ~7 of 16 -- the splat's { alpha >= 1/255 } region inside a 4x4 block, tools/pair_stats.py)")


if __name__ == "__main__":
    main()
