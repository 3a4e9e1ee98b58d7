#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in nefes_amd/libnefes_hip.so (or the library given), from the code objects' notes.
    python tools/kernel_resources.py [lib.so] [name filter]"""
import os, re, subprocess, sys, tempfile

BIN = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([f"{BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
    for i, a in enumerate(starts):
        piece, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"c{i}.o")
        open(piece, "wb").write(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        subprocess.check_call([f"{BIN}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        if os.path.getsize(co):
            yield co


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(root, "nefes_amd", "libnefes_hip.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            notes = subprocess.run([f"{BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                if flt and flt not in name:
                    continue
                agpr = blk.split()[0]
                print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} agpr {agpr:>4s} sgpr {g('sgpr_count'):>4s} "
                      f"spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")


if __name__ == "__main__":
    main()
