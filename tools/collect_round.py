#!/usr/bin/env python3
"""After `gpurun -- bash tools/regen_round.sh <tag>`: condenses gpurun_out/prof_<tag>*/ into profiles/<tag>*_{kernel_stats.csv,pmc.json}
(tools/pmc_summary.py) and copies the measurement files of gpurun_out/<tag>/ to profiles/<tag>_<name>."""
import glob, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for t in (tag, tag + "_batch", tag + "_warm", tag + "_b15", tag + "_b14", tag + "_s32", tag + "_1gib", tag + "_7gib", tag + "_sharded", tag + "_grouped_100mb", tag + "_dealt_100mb"):
    if os.path.isdir(os.path.join(root, "gpurun_out", "prof_" + t)):
        subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), t], stdout=subprocess.DEVNULL, check=True)
for f in glob.glob(os.path.join(root, "gpurun_out", tag, "*")):
    name = os.path.basename(f)
    if name.endswith(".err") or os.path.getsize(f) == 0:
        continue
    shutil.copy(f, os.path.join(root, "profiles", f"{tag}_{name}"))
print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(root, "profiles", tag + "*"))))
