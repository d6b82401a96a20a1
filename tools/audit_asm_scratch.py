"""Fails if a PRODUCTION instantiation of a fused kernel uses scratch memory (a scratch reload waits vmcnt(0), i.e.
drains the weight DMA: 20-40 % slower and, with in-flight asm reads, wrong).  The TAPS debug instantiations (launched
only when a dump is asked for) may spill.   usage: audit_asm_scratch.py file.s [kernel-name substring]"""
import re, sys

# position of the TAPS flag among the bool template arguments (Lb0E / Lb1E in the mangled name); None = no debug form
TAPS_BOOL = {"eval16r_kernel": 1, "eval16_kernel": 1, "evalc_kernel": 1, "eval32_kernel": None,
             "ray_records_kernel": None, "ray_records_c_kernel": None}


def main(path, only=None):
    """only: a substring of the kernel names to audit (pg_train.hip: the persistent layer GEMM counts its own memory
    operations for a counted vmcnt wait, so a compiler-made scratch access would break it; its other kernels may spill)"""
    text = open(path).read()
    bad = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        sym, body = m.group(1), m.group(2)
        if only and only not in sym:
            continue
        size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        kind = next((k for k in TAPS_BOOL if k in sym), sym)
        bools = re.findall(r"Lb([01])E", sym)
        pos = TAPS_BOOL.get(kind)
        debug = pos is not None and len(bools) > pos and bools[pos] == "1"
        dem = sym
        status = "debug instantiation" if debug else ("OK" if size == 0 else "SCRATCH IN A PRODUCTION KERNEL")
        print(f"{dem[:90]:90s} scratch {size:5d} B  {status}")
        if size and not debug:
            bad += 1
    print("SCRATCH AUDIT", "OK" if not bad else "FAILED", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
