#!/usr/bin/env python
"""Turn the two PMC summaries (tools/pmc_summary.py output of separate FETCH_SIZE / WRITE_SIZE passes) into
profiles/r01_pmc_traffic.json: HBM-side bytes per launch of the conv igemm kernels, keyed by bench.py's kernel labels.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B, see
/opt/skills/guides/MI355X_MICROARCH.md "HBM").  Usage: pmc_to_json.py FETCH.txt WRITE.txt OUT.json"""
import hashlib
import json
import os
import re
import sys


def kernel_source_hash():
    """sha256 over the kernel sources (megreader_amd/csrc/*.hip, *.h): bench.py recomputes it and reports `traffic: null`
    when the committed PMC file was measured on other kernels (same function there: keep the two in sync)."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "megreader_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(here)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(here, name), "rb").read())
    return h.hexdigest()[:16]


def parse(path):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        else:
            m = re.search(r"(\w+)\s+n=\s*(\d+)\s+mean=([0-9.e+]+)", line)
            if m and name:
                out[name] = (float(m.group(3)), int(m.group(2)))
    return out


def label(name):
    m = re.search(r"igemm_nt_glds_kernelIDF16bLi(\d+)ELi(\d+)ELi[23]E", name)
    if m:
        return "igemm_nt_kernel<bf16,%s,%s,conv>" % (m.group(1), m.group(2))
    m = re.search(r"igemm_nt_big_kernelIDF16bLi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi[23]E", name)
    if m:   # 8-wave kernels: WM x WN waves of TM x TN MFMA tiles -> a (WM*TM*16) x (WN*TN*16) tile (bench.py labels launches by tile)
        wm, wn, tm, tn = (int(x) for x in m.groups())
        return "igemm_nt_kernel<bf16,%d,%d,conv>" % (wm * tm * 16, wn * tn * 16)
    if re.search(r"igemm_tn_glds_kernel<[12]", name) or re.search(r"igemm_tn_glds_kernelILi[12]E", name):
        return "igemm_tn_kernel<bf16,conv>"
    if "igemm_tn_taps_kernel" in name:
        return "igemm_tn_taps_kernel<bf16,3x3>"
    if "igemm_tn_glds_grouped_kernel" in name:
        return "igemm_tn_glds_grouped_kernel<bf16>"
    m = re.search(r"igemm_nt32_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi[02]E", name)
    if m:   # ping-pong kernel: WM x WN waves of TM x TN 32x32 blocks
        wm, wn, tm, tn = (int(x) for x in m.groups())
        return "igemm_nt_kernel<bf16,%d,%d,conv>" % (wm * tm * 32, wn * tn * 32)
    for fam in ("dcn2_dx_gcol_kernel", "dcn2_coord_gcol_kernel", "dcn2_im2col_kernel", "dcn2_fwd_fused_kernel",
                "dcn2_dx_fused_kernel", "dcn2_coord_fused_kernel", "dcn2_wgrad_fused_kernel"):
        if fam in name:     # DCNv2 kernels (bench.py labels them by family: the phase timer brackets them inside mr_dcn2_fwd / bwd2)
            return fam
    return None


def main():
    fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
    acc = {}
    for name, (fkb, n) in fetch.items():
        lab = label(name)
        if lab is None or name not in write:
            continue
        a = acc.setdefault(lab, [0.0, 0.0, 0])   # launch-weighted sums over every instantiation that maps to the label
        a[0] += fkb * n
        a[1] += write[name][0] * n
        a[2] += n
    out = {}
    for lab, (fsum, wsum, n) in acc.items():
        fkb, wkb = fsum / n, wsum / n
        out[lab] = {"fetch_size_kb_mean": fkb, "write_size_kb_mean": wkb, "launches": n,
                    "bytes_per_launch": 2.0 * fkb * 1024 + wkb * 1024,
                    "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes, mean per launch"}
    out["_kernel_source_hash"] = kernel_source_hash()
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
