#!/usr/bin/env python
"""Development: VGPR count and scratch bytes of every kernel in one csrc/*.hip (device-only compile to ISA text).
    python tools/kernel_regs.py gp_eval_f16.hip [-DSCASML_GP_ABLATE=1]     (also leaves /tmp/<name>.s for reading)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "scasml_gp_amd", "csrc", sys.argv[1])
out = "/tmp/" + sys.argv[1].replace(".hip", ".s")
subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-w", "-S",
                "--cuda-device-only", "-I" + os.path.join(root, "include"), "-o", out, src] + sys.argv[2:], check=True)
t = open(out).read()
# scratch (spill) instructions inside loops, per kernel: the GP kernels count LDS-DMA completions with s_waitcnt
# vmcnt(N), which a spill or reload inside the tile loop would silently break
in_loop = {}
cur, loop = None, False
for line in t.split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur, loop = m.group(1), False
    elif re.match(r"^\.LBB\d+_\d+:", line):
        loop = "Loop" in line
    elif cur and loop and line.strip().startswith("scratch_"):
        in_loop[cur] = in_loop.get(cur, 0) + 1
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    print("%-90s scratch %4s  vgpr %3s%s" % (name[:90], m.group(2), m.group(3),
                                             "   !! %d scratch ops inside loops" % in_loop[m.group(1)] if m.group(1) in in_loop else ""))
