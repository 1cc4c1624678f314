"""Summarise -Rpass-analysis=kernel-resource-usage remarks of one hipcc compile (VGPRs, scratch, occupancy, LDS per kernel).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c X.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> remarks.txt
    python tools/kernel_resources.py remarks.txt [substring filter]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = txt.split("Function Name: ")[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = re.sub(r"^void phihip::", "", d)
        d = re.sub(r"\(.*$", "", d)
        if flt and flt not in d:
            continue
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"vgpr={g('VGPRs'):3d} agpr={g('AGPRs'):3d} scratch={scratch:4d} occ={occ} lds={lds:6d} sgpr={g('SGPRs'):3d} sgpr_spill={g('SGPRs Spill'):4d} "
              f"vgpr_spill={g('VGPRs Spill'):3d}  {d}")


if __name__ == "__main__":
    main()
