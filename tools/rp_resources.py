#!/usr/bin/env python3
"""Print VGPR / scratch use of every gemv_repacked_kernel<G, T, W, NT, MT, EPI> instantiation (hipcc resource remarks).

The launch heuristic's `rp_fits` table in sglang_awq_amd/csrc/awq_repacked.hip is derived from this output;
run it after touching the kernel.  Exit status 1 if any built instantiation uses scratch.
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "sglang_awq_amd", "csrc", f) for f in ("awq_repacked.hip", "awq_repacked_fused.hip", "awq_repacked_loop.hip", "awq_repacked_ext.hip",
                                                                   "awq_repacked_prefill.hip")]


def main():
    txt = ""
    with tempfile.TemporaryDirectory() as d:
        for src in SRCS:
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=on",
                   "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "x.o")]
            txt += subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = blk.split()[0]
        m = re.search(r"gemv_repacked_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELi(\d+)E", name)
        if not m:
            continue
        G, T, W, NT, MT, EPI = (int(v) for v in m.groups())
        PRO = 0

        def field(k):
            mm = re.search(k + r": (\d+)", blk)
            return int(mm.group(1)) if mm else -1

        rows.append((PRO, EPI, MT, W, NT, G, T, field("VGPRs"), field(r"ScratchSize \[bytes/lane\]"), field(r"Occupancy \[waves/SIMD\]")))
    rows.sort()
    bad = 0
    for r in rows:
        print("PRO%-2d EPI%d MT%d W%-2d NT%d G%d T%d  vgpr %3d  scratch %4d  occupancy %d" % r)
        bad += r[8] > 0
    print(f"{len(rows)} gemv_repacked_kernel instantiations, {bad} with scratch")
    # every other kernel of these files (gemv_rp2_kernel, gemv_rpx_kernel, the prefill kernels, repack): scratch must be 0 too
    other, other_bad, worst = 0, [], {}
    for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = blk.split()[0]
        if "gemv_repacked_kernelI" in name:
            continue
        mm = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk)
        vg = re.search(r" VGPRs: (\d+)", blk)
        other += 1
        fam = re.sub(r"I.*", "", name.replace("_ZN3awq", ""))
        worst[fam] = max(worst.get(fam, 0), int(vg.group(1)) if vg else 0)
        if mm and int(mm.group(1)) > 0:
            other_bad.append(name)
    print(f"{other} other kernels, {len(other_bad)} with scratch; max VGPRs per family: " + ", ".join(f"{k} {v}" for k, v in sorted(worst.items())))
    for n in other_bad[:20]:
        print("  scratch:", n)
    return 1 if (bad or other_bad) else 0


if __name__ == "__main__":
    sys.exit(main())
