#!/usr/bin/env python3
"""Disassembly of one kernel out of libnefes_hip.so or an object file of the build (anything with a .hip_fatbin section), and its
instruction mix.   python tools/disasm_kernel.py <lib.so | file.o> <substring of the mangled kernel name> [out.s]"""
import collections, os, re, subprocess, sys, tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as K


def kernel_body(path, key):
    with tempfile.TemporaryDirectory() as tmp:
        for co in K.code_objects(path, tmp):
            sym = subprocess.run([f"{K.BIN}/llvm-readelf", "-s", co], capture_output=True, text=True).stdout
            if key not in sym:
                continue
            lines = subprocess.run([f"{K.BIN}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.split("\n")
            starts = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)]
            for n, i in enumerate(starts):
                if key in lines[i]:
                    return lines[i], lines[i + 1:starts[n + 1] if n + 1 < len(starts) else len(lines)]
    raise SystemExit(f"no kernel matching {key!r} in {path}")


def main():
    head, body = kernel_body(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(head + "\n" + "\n".join(body))
    ops = collections.Counter(m.group(1) for m in (re.match(r"\s+(\S+)", l) for l in body) if m)
    mfma = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
    valu = sum(v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(head.strip())
    print(f"{sum(ops.values())} instructions: {mfma} MFMA, {valu} other vector ({valu / max(mfma, 1):.2f} per MFMA), "
          f"{sum(v for k, v in ops.items() if k.startswith('ds_'))} LDS, {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_')))} global, "
          f"{ops['s_waitcnt']} s_waitcnt, {ops['s_nop']} s_nop, {ops['s_barrier']} s_barrier")
    for k, v in ops.most_common(30):
        print(f"  {k:28s} {v:6d}  {v / max(mfma, 1):.3f} per MFMA")


if __name__ == "__main__":
    main()
