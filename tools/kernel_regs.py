"""Print scratch/SGPR/VGPR usage of kernels in build/ivs_api.s (made by `make -C iv_interpolation_amd/csrc asm`)."""
import re
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else ""
t = open("build/ivs_api.s").read()
rx = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?"
                r"\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)")
for m in rx.finditer(t):
    if re.search(pat, m.group(1)):
        print(m.group(1), "scratch", m.group(2), "sgpr", m.group(3), "vgpr", m.group(4))
