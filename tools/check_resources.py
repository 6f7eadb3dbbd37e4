"""List every kernel's VGPRs / scratch / occupancy (hipcc -Rpass-analysis=kernel-resource-usage); flag spills."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bio_image_unet_amd", "csrc")
only_spills = "--all" not in sys.argv
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                          "-I", CSRC, "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    name, d = None, {}
    for l in out.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            name, d = m.group(1), {}
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", l)
        if m and name:
            d[m.group(1).split()[0]] = int(m.group(2))
        if "LDS Size" in l and name:
            if not only_spills or d.get("ScratchSize", 0) > 0:
                print(f"{os.path.basename(src):22s} {name[:70]:70s} vgpr={d.get('VGPRs')} scratch={d.get('ScratchSize')} occ={d.get('Occupancy')}")
            name = None
