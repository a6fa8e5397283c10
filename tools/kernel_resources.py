"""Print a table of per-kernel register / LDS / scratch usage for the gfx950 build.
usage: python tools/kernel_resources.py [file.hip ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "cpc_audio_amd", "csrc", "*.hip")))
for src in srcs:
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    print(f"== {os.path.basename(src)}")
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        print(f"  {name[:60]:60s} vgpr={v.get('VGPRs',0):3d} agpr={v.get('AGPRs',0):3d} sgpr={v.get('TotalSGPRs',0):3d} "
              f"scratch={v.get('ScratchSize',0):4d} spill={v.get('VGPRs Spill',0):3d} lds={v.get('LDS Size',0):6d} occ={v.get('Occupancy',0)}")
