"""Compact per-kernel resource table (VGPRs, AGPRs, spills, scratch, LDS) of one .hip file, from hipcc's kernel-resource-usage remarks.

    python tools/kernel_resources.py bio_image_unet_amd/csrc/biu_conv_mfma.hip [filter] [-DFLAG ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-I", os.path.join(ROOT, "include"), "-I",
       os.path.join(ROOT, "bio_image_unet_amd", "csrc"), "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v; rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, r in rows.items():
    dm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else name
    short = re.sub(r"\(anonymous namespace\)::|__hip_bfloat16|void ", "", dm)[:90]
    if flt and flt not in short:
        continue
    print(f"{short:<92} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} vspill {r.get('VGPRs Spill','?'):>3} sspill {r.get('SGPRs Spill','?'):>3} scratch {r.get('ScratchSize','?'):>4} occ {r.get('Occupancy','?')}")
