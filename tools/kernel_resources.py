"""Per-kernel register / LDS / scratch usage of the built libpdengine.so (runs on the CPU box: reads the code objects).

    python tools/kernel_resources.py [filter-substring]

Prints name, VGPRs, AGPRs, SGPRs, static LDS, scratch bytes (non-zero scratch = spills)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "prompt-diffusion_amd", "csrc", "libpdengine.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, os.path.join(d, "x.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(MAGIC, blob)] + [len(blob)]
        rows = []
        for i in range(len(starts) - 1):
            part = os.path.join(d, f"b{i}.bin")
            open(part, "wb").write(blob[starts[i]:starts[i + 1]])
            co = os.path.join(d, f"b{i}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode or not os.path.exists(co):
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "agpr_count" and cur.get("name"):
                    rows.append(cur)
                    cur = {}
                if k in ("name", "vgpr_count", "sgpr_count", "agpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
                    cur[k] = v
            if cur.get("name"):
                rows.append(cur)
        seen = set()
        for r in rows:
            n = r.get("name", "")
            if n in seen:
                continue
            seen.add(n)
            dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(GemmParams\)|\(AttnParams\)|\(anonymous namespace\)::", "", dem)
            if flt not in dem and flt not in n:
                continue
            print(f"{dem[:110]:110s} vgpr {r.get('vgpr_count', '?'):>4s} agpr {r.get('agpr_count', '?'):>3s} sgpr {r.get('sgpr_count', '?'):>3s} "
                  f"lds {r.get('group_segment_fixed_size', '?'):>6s} scratch {r.get('private_segment_fixed_size', '?')}")


if __name__ == "__main__":
    main()
