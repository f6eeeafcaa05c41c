#!/usr/bin/env python3
"""Reading aid (here, no GPU): compile every kernel source for gfx950 with -save-temps and print, per kernel, what the compiler
did with its memory operations -- spills (scratch_*), loads, the s_waitcnt vmcnt it inserted (and how many of them are
vmcnt(0)), and the number of load -> wait alternations ("serial round trips"; a kernel that is meant to have N loads in
flight and shows N alternations runs them one after the other).   python scripts/experiments/isa_scan.py [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
files = sys.argv[1:] or sorted(glob.glob(f"{R}/mbexwn_vocoder_amd/csrc/*.hip"))
tmp = tempfile.mkdtemp()
for f in files:
    base = os.path.basename(f)[:-4]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", f, "-o", f"{tmp}/{base}.o", "-save-temps=obj"],
                   check=True, stderr=subprocess.DEVNULL, cwd=tmp)
    asm = open(f"{tmp}/{base}-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    for m in re.finditer(r"^(_Z\w+):.*?^\.Lfunc_end", asm, re.S | re.M):
        body = m.group(0)
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)[:60]
        ops = re.findall(r"^\s+(global_load\w*|global_store\w*|scratch_\w+|s_waitcnt vmcnt\(\d+\)|v_mfma\w+|s_barrier)", body, re.M)
        loads = sum(o.startswith("global_load") and "lds" not in o for o in ops)
        dma = sum("load_lds" in o for o in ops)
        waits = [o for o in ops if o.startswith("s_waitcnt")]
        w0 = sum(o.endswith("vmcnt(0)") for o in waits)
        scratch = sum(o.startswith("scratch") for o in ops)
        mfma = sum(o.startswith("v_mfma") for o in ops)
        # alternations: a wait directly preceded (in this filtered stream) by a load
        alt = sum(1 for a, b in zip(ops, ops[1:]) if a.startswith("global_load") and "lds" not in a and b.startswith("s_waitcnt"))
        print(f"{base:18s} {name:60s} loads {loads:3d} lds-dma {dma:3d} waits {len(waits):3d} (vmcnt(0): {w0:3d}) load->wait {alt:3d} scratch {scratch:3d} mfma {mfma:4d}")
