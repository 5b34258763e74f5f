"""Registers, scratch, LDS and the compiler's occupancy figure of every kernel of the library (hipcc -S, no GPU needed), with a flag
where the scalar registers alone cap a SIMD at seven waves (more than 96: a wave allocates them in blocks of 16, a SIMD has 800) -- how
round 4 found the two-workgroups-per-CU instances of ordered_lean_kernel running one.   usage: kernel_resources.py [file.hip ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dither_pie_amd", "csrc")
files = sys.argv[1:] or ["ordered.hip", "accel.hip", "ediff.hip", "vardiff.hip", "kmeans.hip", "kmeans_hist.hip", "distinct.hip", "bluenoise.hip"]
flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function --cuda-device-only -S".split()
for f in files:
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if f == "kmeans.hip" else []
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags, *extra, f, "-o", out], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        name, rec = None, {}
        for line in open(out):
            m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
            if m:
                name, rec = m.group(1), {}
            for key in ("TotalNumSgprs", "NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize", "SGPRBlocks", "VGPRBlocks"):
                m2 = re.match(r"; %s: (\d+)" % key, line)
                if m2:
                    rec[key] = int(m2.group(1))
            if line.startswith("; Occupancy") and name:
                d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                d = d.replace("void dp::(anonymous namespace)::", "").replace("dp::(anonymous namespace)::", "")
                d = re.sub(r"\(unsigned.*|\(int.*|\(dp::.*|\(double.*|\(float.*", "", d)
                sg, lds = 8 * (rec.get("SGPRBlocks", 0) + 1), rec.get("LDSByteSize", 0)   # what a wave ALLOCATES
                per_cu = 163840 // lds if lds else 0
                flag = "  <-- more than 96 SGPRs: at most 7 waves per SIMD" if sg > 96 and 2 <= per_cu else ""
                print(f"{f:16s} {d[:70]:70s} sgpr {rec.get('TotalNumSgprs', 0):3d} (allocated {sg:3d}) vgpr {rec.get('NumVgprs', 0):3d} (allocated {8 * (rec.get('VGPRBlocks', 0) + 1):3d}) scratch {rec.get('ScratchSize', 0):5d} "
                      f"lds {lds:6d} ({per_cu or '-'} per CU by LDS) occupancy {rec.get('Occupancy', 0)}{flag}", flush=True)
                name = None
