#!/usr/bin/env python3
"""Software wait-state linter for the gfx950 code objects of libnefes_hip.so.

hipcc's hazard recognizer pads the "manually inserted wait states" of the CDNA3/4 ISA only between instructions it can see: an
`asm volatile` statement is one opaque instruction to it (it is neither a VALU nor an MFMA to GCNHazardRecognizer, and
`getWaitStatesSince` skips it), so every dependency whose producer OR consumer sits inside inline asm is the program's to pad --
and the field kernels of this library are built from such statements (csrc/field_h3.h, field_common.h, field_x6.h).  This tool
disassembles every kernel and re-checks, instruction by instruction and across basic blocks, the rules LLVM applies on
gfx940 / gfx950 (GCNHazardRecognizer: checkMAIHazards90A, checkMAIVALUHazards, checkVALUHazards, checkPermlaneHazards,
checkReadM0Hazards, checkVMEM / DPP / div_fmas / rw-lane hazards), on the FINAL instruction stream, where asm and compiler code
are the same thing.  One instruction = one wait state, `s_nop N` = N + 1 (LLVM's own count).

Rules (R = wait states required between producer and consumer; N = passes of the producing MFMA: 32x32x16 f16 / bf16 = 8,
16x16x32 = 4, 32x32x2 f32 = 16; XDL = every MFMA but the f32 / f64 ones):

    A1  VALU write of a VGPR / AGPR (v_accvgpr_write included)  -> MFMA reads it as SrcA / SrcB / SrcC              R = 2
    A2  MFMA write -> MFMA reads an OVERLAPPING (not identical) range as SrcC:   XDL -> XDL / SGEMM  N + 2;  SGEMM -> SGEMM  N;
        SGEMM -> XDL 0;  identical range (the accumulate chain): 0 (2 after a 2-pass producer)
    A3  MFMA write -> MFMA reads it as SrcA / SrcB:                              XDL  N + 3 + (N != 2);   SGEMM  N + 2
    A4  VALU write of EXEC -> MFMA                                                                                      R = 4
    B1  MFMA write -> VALU / VMEM / LDS / FLAT reads it:                         XDL  N + 3 + (N != 2);   SGEMM  N + 2
    B2  MFMA write -> VALU (or a load) writes it (write after write):            same as B1
    B3  XDL MFMA reads SrcC -> VALU writes that register (write after read):     N = 2: 1, 4: 3, 8: 7, 16: 15
    C1  VALU write of an SGPR -> VMEM reads it (address, offset, descriptor)                                            R = 5
    C2  VALU write of an SGPR -> v_readlane / v_writelane lane select                                                   R = 4
    C3  VALU write of VCC -> v_div_fmas                                                                                 R = 4
    C4  VALU write of EXEC -> DPP instruction  R = 5;  VALU write of a VGPR -> DPP instruction reads it                 R = 2
    C5  SALU write of M0 -> LDS-DMA (global_load_lds_*), s_movrel, GWS, s_sendmsg                                       R = 1
    C6  store of more than 64 bits of data (global / flat / scratch dwordx3 / x4) -> VALU writes its data registers     R = 2
    C7  transcendental VALU (v_exp, v_log, v_rcp, v_rsq, v_sqrt, v_sin, v_cos) -> non-transcendental VALU reads it      R = 1
    C8  SDWA write with dst_sel != DWORD -> VALU reads it                                                              R = 1
    C9  VALU write of an operand of v_permlane{16,32}_swap -> the swap                                                  R = 2
    C10 VALU write of an SGPR / VCC -> VALU reads it as a scalar operand (gfx940 "VALU / decoder co-execution")          R = 2
    C11 VALU write of a VGPR -> v_readlane / v_readfirstlane reads it  R = 1;  VALU write of EXEC -> v_read*lane / v_writelane  R = 4

What MI355X itself requires, measured (csrc/hazard_probe.hip): B1 = 5 after an XDL MFMA (4 for an LDS store)
and nothing after the fp32 MFMA (interlocked), B2 = 8, A1 = 1 for SrcA / SrcB and nothing for SrcC, A3 / B3 / C10 interlocked.  The rules
below stay at LLVM's numbers: a superset, and what compiler-placed code is padded to.

    python tools/hazard_lint.py [lib.so | file.o | file.s] [kernel-name filter]      exit status 1 when a rule is violated
"""
import collections, os, re, subprocess, sys, tempfile

BIN = "/opt/rocm/lib/llvm/bin"
MAXWS = 21                    # no rule looks further back than this many wait states

# ---- registers ---------------------------------------------------------------------------------------------------------------
VCC, EXEC, M0 = ("s", 106), ("s", 126), ("s", 124)
_SPECIAL = {"vcc": (VCC, 2), "vcc_lo": (VCC, 1), "vcc_hi": (("s", 107), 1), "exec": (EXEC, 2), "exec_lo": (EXEC, 1),
            "exec_hi": (("s", 127), 1), "m0": (M0, 1), "flat_scratch": (("s", 102), 2), "flat_scratch_lo": (("s", 102), 1),
            "flat_scratch_hi": (("s", 103), 1), "xnack_mask": (("s", 104), 2)}
_REG = re.compile(r"^([vas])(\d+)$")
_RANGE = re.compile(r"^([vas])\[(\d+):(\d+)\]$")
_TTMP = re.compile(r"^ttmp(\d+)$")
_TTMPR = re.compile(r"^ttmp\[(\d+):(\d+)\]$")


def parse_reg(tok):
    """operand text -> (file, first index, count) or None for literals / modifiers / labels"""
    t = tok.strip()
    t = re.sub(r"^(-|neg\(|abs\(|sext\()+", "", t)
    t = t.strip("|)")
    t = re.sub(r"^-", "", t).strip("|")
    m = _REG.match(t)
    if m:
        return (m.group(1), int(m.group(2)), 1)
    m = _RANGE.match(t)
    if m:
        return (m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
    if t in _SPECIAL:
        (f, i), n = _SPECIAL[t]
        return (f, i, n)
    m = _TTMP.match(t)
    if m:
        return ("s", 108 + int(m.group(1)), 1)
    m = _TTMPR.match(t)
    if m:
        return ("s", 108 + int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1)
    return None


def split_operands(text):
    """'v[10:11], v[12:13], v[54:55] op_sel:[0,1]' -> (['v[10:11]', 'v[12:13]', 'v[54:55]'], 'op_sel:[0,1]')"""
    ops, depth, cur = [], 0, ""
    for ch in text:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    mods = ""
    clean = []
    for i, o in enumerate(ops):
        parts = o.split(None, 1)
        # modifiers ride behind the last operand, separated by blanks ("v1 offset:16 nt", "v[54:55] op_sel:[0,1]")
        if len(parts) == 2 and not parts[0].endswith(","):
            clean.append(parts[0])
            mods += " " + parts[1]
        else:
            clean.append(o)
    # an operand list may consist of modifiers only ("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return clean, mods.strip()


# ---- instruction model -----------------------------------------------------------------------------------------------------
MFMA_PASSES = {          # gfx950 (MI355X_MICROARCH.md: 32x32x16 = 32 cycles = 8 passes of 4; 16x16x32 = 16 cycles; 32x32x2 f32 = 64 cycles)
    "v_mfma_f32_32x32x16_f16": (8, True), "v_mfma_f32_32x32x16_bf16": (8, True),
    "v_mfma_f32_16x16x32_f16": (4, True), "v_mfma_f32_16x16x32_bf16": (4, True),
    "v_mfma_f32_32x32x2_f32": (16, False), "v_mfma_f32_16x16x4_f32": (8, False), "v_mfma_f32_4x4x1_16b_f32": (2, False),
    "v_mfma_f32_32x32x1_2b_f32": (16, False), "v_mfma_f32_16x16x1_4b_f32": (8, False),
    "v_mfma_f32_32x32x8_f16": (16, True), "v_mfma_f32_16x16x16_f16": (8, True),
    "v_mfma_f32_32x32x8_bf16_1k": (16, True), "v_mfma_f32_16x16x16_bf16_1k": (8, True),
    "v_mfma_f32_32x32x64_f8f6f4": (16, True), "v_mfma_f32_16x16x128_f8f6f4": (8, True),
    "v_mfma_i32_32x32x32_i8": (8, True), "v_mfma_i32_16x16x64_i8": (4, True),
}
TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|f64|legacy_f32)")
BRANCH = re.compile(r"^s_(branch|cbranch_\w+|setpc_b64|swappc_b64|endpgm|trap)")
TWO_DST = re.compile(r"^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co)_u32|^v_div_scale_|^v_mad_(u64_u32|i64_i32)")
NO_DST_SALU = re.compile(r"^s_(cmp|cmpk|bitcmp|waitcnt|nop|barrier|sleep|sethalt|setprio|sendmsg|icache_inv|dcache|ttrace|"
                         r"endpgm|branch|cbranch|setpc|trap|incperflevel|decperflevel|set_gpr_idx|setvskip|code_end|rfe)")


class Inst:
    __slots__ = ("addr", "mnem", "ops", "mods", "text", "kind", "defs", "uses", "srcc", "srcab", "dst", "passes", "xdl", "ws",
                 "target", "trans", "dpp", "sdwa_part", "big_store_data", "lane_sel", "vmem_sgprs", "scalar_uses", "is_branch",
                 "is_end", "reads_m0_dma", "vgpr_src0")


def _regs(r):
    return [(r[0], r[1] + k) for k in range(r[2])] if r else []


def decode(addr, mnem, optext, comment):
    I = Inst()
    I.addr, I.mnem, I.text = addr, mnem, (mnem + " " + optext).strip()
    ops, mods = split_operands(optext)
    I.ops, I.mods = ops, mods
    I.defs, I.uses, I.srcc, I.srcab, I.dst = [], [], None, [], None
    I.passes, I.xdl, I.ws, I.target = 0, False, 1, None
    I.trans = I.dpp = I.sdwa_part = I.is_branch = I.is_end = I.reads_m0_dma = False
    I.big_store_data, I.lane_sel, I.vmem_sgprs, I.scalar_uses, I.vgpr_src0 = [], [], [], [], []
    R = [parse_reg(o) for o in ops]

    def use(r):
        I.uses.extend(_regs(r))

    def df(r):
        I.defs.extend(_regs(r))

    if mnem.startswith("v_mfma") or mnem.startswith("v_smfmac"):
        if mnem not in MFMA_PASSES:
            raise SystemExit(f"hazard_lint: no pass count for {mnem}")
        I.kind = "mfma"
        I.passes, I.xdl = MFMA_PASSES[mnem]
        I.dst = R[0]
        df(R[0])
        I.srcab = [r for r in R[1:3] if r]
        for r in R[1:3]:
            use(r)
        if len(R) > 3 and R[3]:
            I.srcc = R[3]
            use(R[3])
        return I
    if mnem.startswith("v_"):
        I.kind = "valu"
        I.trans = bool(TRANS.match(mnem))
        I.dpp = "_dpp" in mnem or "row_" in mods or "quad_perm" in mods or "wave_" in mods
        if "dst_sel:" in mods and "dst_sel:DWORD" not in mods:
            I.sdwa_part = True
        nd = 1
        if TWO_DST.match(mnem):
            nd = 2
        if mnem.startswith("v_cmpx"):
            I.defs.extend([EXEC, ("s", 127)])
            nd = 1 if (R and R[0] and R[0][0] == "s") else 0
        if mnem.startswith(("v_permlane32_swap", "v_permlane16_swap", "v_swap_b")):
            nd = 2
            for r in R[:2]:
                use(r)
        for r in R[:nd]:
            df(r)
        for r in R[nd:]:
            use(r)
        # read-modify-write destinations
        if re.match(r"^v_(fmac|mac|pk_fmac|dot\w*c|fma_mixlo|fma_mixhi|mad_mixlo|mad_mixhi|writelane|cvt_scalef32_pk)", mnem) or I.dpp or I.sdwa_part:
            use(R[0])
        if mnem.startswith(("v_div_fmas", "v_addc_co", "v_subb_co", "v_subbrev_co")) and VCC not in I.uses and not any(
                r and r[0] == "s" for r in R[2:]):
            I.uses.extend([VCC, ("s", 107)])
        if mnem.startswith(("v_readlane", "v_writelane")) and len(R) > 2 and R[2] and R[2][0] == "s":
            I.lane_sel = _regs(R[2])
        if mnem.startswith(("v_readlane", "v_readfirstlane")) and len(R) > 1 and R[1] and R[1][0] in "va":
            I.vgpr_src0 = _regs(R[1])
        I.scalar_uses = [u for u in I.uses if u[0] == "s" and u not in (EXEC, ("s", 127))]
        return I
    if mnem.startswith(("global_", "flat_", "scratch_", "buffer_", "tbuffer_", "image_")):
        I.kind = "vmem"
        is_store = "_store" in mnem
        is_lds = "_lds_" in mnem or mnem.endswith("_lds") or " lds" in (" " + mods)
        is_atomic = "atomic" in mnem
        if is_store or is_lds or (is_atomic and "sc0" not in mods and "glc" not in mods):
            for r in R:
                use(r)
        else:
            df(R[0])
            for r in R[1:]:
                use(r)
        if is_lds:
            I.reads_m0_dma = True
            I.uses.append(M0)
        if is_store and re.search(r"dwordx[34]$", mnem):
            # global / flat / scratch: data is the second operand; buffer stores: the first (hazard only with an SGPR soffset: LLVM)
            if mnem.startswith(("global_", "flat_", "scratch_")):
                I.big_store_data = _regs(R[1]) if len(R) > 1 else []
            elif len(R) > 3 and R[3] and R[3][0] == "s":
                I.big_store_data = _regs(R[0])
        I.vmem_sgprs = [u for r in R for u in _regs(r) if r and r[0] == "s"]
        return I
    if mnem.startswith("ds_"):
        I.kind = "ds"
        if re.match(r"^ds_(read|bpermute|permute|swizzle|consume|append|ordered|.*_rtn)", mnem):
            df(R[0])
            for r in R[1:]:
                use(r)
        else:
            for r in R:
                use(r)
        if mnem.startswith("ds_gws"):
            I.reads_m0_dma = True
        return I
    if mnem.startswith("s_"):
        I.is_branch = bool(BRANCH.match(mnem))
        I.is_end = mnem in ("s_endpgm", "s_setpc_b64", "s_branch", "s_trap")
        if mnem == "s_nop":
            I.kind = "nop"
            I.ws = int(ops[0], 0) + 1 if ops else 1
            return I
        I.kind = "smem" if mnem.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache",
                                            "s_scratch_load", "s_atc")) else "salu"
        if I.is_branch:
            m = re.search(r"<.*\+0x([0-9a-f]+)>\s*$", comment)
            if m and mnem not in ("s_setpc_b64", "s_swappc_b64"):
                I.target = int(m.group(1), 16)
            for r in R:
                use(r)
            return I
        if NO_DST_SALU.match(mnem):
            for r in R:
                use(r)
        else:
            if R:
                df(R[0])
            for r in R[1:]:
                use(r)
            if "saveexec" in mnem:
                I.defs.extend([EXEC, ("s", 127)])
                I.uses.extend([EXEC, ("s", 127)])
        if mnem.startswith(("s_movrel", "s_sendmsg")):
            I.reads_m0_dma = True
        return I
    I.kind = "other"
    return I


# ---- disassembly ----------------------------------------------------------------------------------------------------------------
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


def disassemble(path, tmp):
    """yields the text of `llvm-objdump -d -C` for every gfx950 code object in a library / host object, or the file itself (.s)"""
    if path.endswith(".s"):
        yield open(path).read()
        return
    head = open(path, "rb").read(64)
    is_device_elf = head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00"          # e_machine = EM_AMDGPU (224)
    cos = [path] if is_device_elf else list(code_objects(path, tmp))
    for co in cos:
        yield subprocess.run([f"{BIN}/llvm-objdump", "-d", "-C", "--no-show-raw-insn", co], capture_output=True, text=True,
                             check=True).stdout


_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$")
_FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")


def functions(text):
    """-> [(name, base address, [Inst])]"""
    out, cur = [], None
    for line in text.split("\n"):
        m = _FUNC.match(line)
        if m:
            cur = (m.group(2), int(m.group(1), 16), [])
            out.append(cur)
            continue
        if cur is None:
            continue
        m = _LINE.match(line)
        if not m:
            continue
        mnem, optext, addr, rest = m.group(1), m.group(2), int(m.group(3), 16), m.group(4)
        cur[2].append(decode(addr, mnem, optext, rest))
    return out


# ---- the check -----------------------------------------------------------------------------------------------------------------
class W:                      # a recent writer / reader of one register
    __slots__ = ("t", "inst")

    def __init__(self, t, inst):
        self.t, self.inst = t, inst


def mfma_read_ws(p):          # B1 / B2 / A3: wait states after an MFMA before its result may be touched
    return p.passes + 3 + (1 if p.passes != 2 else 0) if p.xdl else p.passes + 2


def srcc_overlap_ws(p, c):    # A2 (gfx950 columns)
    if p.xdl:
        return p.passes + 2 if c.xdl else p.passes + 1 + (1 if p.passes != 2 else 0)
    return 0 if c.xdl else p.passes


WAR_WS = {2: 1, 4: 3, 8: 7, 16: 15}


class State:
    """what the last MAXWS wait states did to each register; t = wait-state clock at which the NEXT instruction would issue"""

    def __init__(self):
        self.t = 0
        self.valu_w = {}        # reg -> W: last non-MFMA VALU write
        self.mfma_w = {}        # reg -> W: last MFMA write
        self.mfma_c = {}        # reg -> W: last XDL MFMA that read it as SrcC
        self.salu_m0 = None     # time of the last SALU write of M0
        self.big_store = {}     # reg -> W: data register of a > 64-bit store
        self.trans_w = {}
        self.sdwa_w = {}

    def copy(self):
        s = State()
        s.t = self.t
        for k in ("valu_w", "mfma_w", "mfma_c", "big_store", "trans_w", "sdwa_w"):
            setattr(s, k, dict(getattr(self, k)))
        s.salu_m0 = self.salu_m0
        return s


def step(S, I, report):
    """check I against S, then record I; report(rule, need, have, producer)"""
    t = S.t

    def since(w):                 # wait states between producer and this instruction
        return t - w.t

    def need(rule, w, n):
        if w is not None and n > 0 and since(w) < n:
            report(rule, n, since(w), w.inst)

    k = I.kind
    if k == "mfma":
        for r in I.uses:
            need("A1", S.valu_w.get(r), 2)
        need("A4", S.valu_w.get(EXEC), 4)
        if I.srcc:
            regs = _regs(I.srcc)
            seen = set()
            for r in regs:
                w = S.mfma_w.get(r)
                if w is None or id(w.inst) in seen:
                    continue
                seen.add(id(w.inst))
                p = w.inst
                if p.dst == I.srcc:
                    need("A2", w, 2 if p.passes == 2 else 0)
                else:
                    need("A2", w, srcc_overlap_ws(p, I))
        for ab in I.srcab:
            for r in _regs(ab):
                w = S.mfma_w.get(r)
                if w:
                    need("A3", w, mfma_read_ws(w.inst))
    elif k in ("valu", "vmem", "ds"):
        for r in I.uses:
            w = S.mfma_w.get(r)
            if w:
                need("B1", w, mfma_read_ws(w.inst))
        for r in I.defs:
            w = S.mfma_w.get(r)
            if w:
                need("B2", w, mfma_read_ws(w.inst))
            if k == "valu":
                w = S.mfma_c.get(r)
                if w:
                    need("B3", w, WAR_WS[w.inst.passes])
                need("C6", S.big_store.get(r), 2)
        if k == "vmem":
            for r in I.vmem_sgprs:
                need("C1", S.valu_w.get(r), 5)
            if I.reads_m0_dma and S.salu_m0 is not None:
                need("C5", S.salu_m0, 1)
        if k == "ds" and I.reads_m0_dma and S.salu_m0 is not None:
            need("C5", S.salu_m0, 1)
        if k == "valu":
            for r in I.lane_sel:
                need("C2", S.valu_w.get(r), 4)
            if I.mnem.startswith("v_div_fmas"):
                need("C3", S.valu_w.get(VCC), 4)
            if I.dpp:
                need("C4", S.valu_w.get(EXEC), 5)
                for r in I.uses:
                    if r[0] in "va":
                        need("C4", S.valu_w.get(r), 2)
            if not I.trans:
                for r in I.uses:
                    need("C7", S.trans_w.get(r), 1)
            for r in I.uses:
                need("C8", S.sdwa_w.get(r), 1)
            if I.mnem.startswith(("v_permlane32_swap", "v_permlane16_swap")):
                for r in I.uses:
                    need("C9", S.valu_w.get(r), 2)
            for r in I.scalar_uses:
                need("C10", S.valu_w.get(r), 2)
            for r in I.vgpr_src0:
                need("C11", S.valu_w.get(r), 1)
            if I.mnem.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                need("C11", S.valu_w.get(EXEC), 4)
    elif k == "salu" and I.reads_m0_dma and S.salu_m0 is not None:
        need("C5", S.salu_m0, 1)

    # ---- record ----
    S.t = t + I.ws
    me = W(S.t, I)              # "since" of the next instruction = 0
    if k == "mfma":
        for r in I.defs:
            S.mfma_w[r] = me
            S.valu_w.pop(r, None)
        if I.xdl and I.srcc:
            for r in _regs(I.srcc):
                S.mfma_c[r] = me
    elif k == "valu":
        for r in I.defs:
            S.valu_w[r] = me
            S.mfma_w.pop(r, None)
            S.mfma_c.pop(r, None)
            S.big_store.pop(r, None)
            if I.trans:
                S.trans_w[r] = me
            else:
                S.trans_w.pop(r, None)
            if I.sdwa_part:
                S.sdwa_w[r] = me
            else:
                S.sdwa_w.pop(r, None)
    else:
        for r in I.defs:        # loads, SALU, SMEM: a later write by anything else ends the VALU / MFMA producer's claim
            S.valu_w.pop(r, None)
            S.mfma_w.pop(r, None)
            S.trans_w.pop(r, None)
            S.sdwa_w.pop(r, None)
        if k == "salu" and M0 in I.defs:
            S.salu_m0 = me
        if k == "vmem" and I.big_store_data:
            for r in I.big_store_data:
                S.big_store[r] = me


def lint_function(name, base, insts):
    """-> [(addr, rule, need, have, producer text, consumer text)]"""
    if not insts:
        return []
    index = {I.addr: n for n, I in enumerate(insts)}
    leaders = {0}
    for n, I in enumerate(insts):
        if I.is_branch:
            if n + 1 < len(insts):
                leaders.add(n + 1)
            if I.target is not None and base + I.target in index:
                leaders.add(index[base + I.target])
    order = sorted(leaders)
    bstart = {b: i for i, b in enumerate(order)}
    bend = {b: (order[i + 1] if i + 1 < len(order) else len(insts)) for i, b in enumerate(order)}
    preds = collections.defaultdict(list)
    for b in order:
        last = insts[bend[b] - 1]
        if last.is_branch and last.target is not None and base + last.target in index:
            preds[index[base + last.target]].append(b)
        if not last.is_end and bend[b] < len(insts):
            preds[bend[b]].append(b)
    found = {}

    def run(seq, first_reported, limit_ws=None):
        S = State()
        ws = 0
        for n, I in enumerate(seq):
            if n >= first_reported:
                def rep(rule, need, have, prod, I=I):
                    found.setdefault((I.addr, rule), (I.addr, rule, need, have, prod.text, I.text))
                step(S, I, rep)
                ws += I.ws
                if limit_ws is not None and ws > limit_ws:
                    break
            else:
                step(S, I, lambda *a: None)

    def tails(b, budget, depth):
        """instruction sequences that may precede block b: the last `budget` wait states along every path"""
        out = []
        for p in preds.get(b, []):
            seq, ws = [], 0
            for I in reversed(insts[p:bend[p]]):
                seq.append(I)
                ws += I.ws
                if ws >= budget:
                    break
            seq.reverse()
            if ws < budget and depth > 0 and preds.get(p):
                for more in tails(p, budget - ws, depth - 1):
                    out.append(more + seq)
            else:
                out.append(seq)
        return out

    for b in order:
        body = insts[b:bend[b]]
        run(body, 0)
        for pre in tails(b, MAXWS, 3):
            run(pre + body, len(pre), MAXWS)
    return sorted(found.values())


def lint(path, name_filter=""):
    """-> (violations [(kernel, addr, rule, need, have, producer, consumer)], kernels checked, instructions checked)"""
    out, nk, ni = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for text in disassemble(path, tmp):
            for name, base, insts in functions(text):
                if name_filter and name_filter not in name:
                    continue
                nk += 1
                ni += len(insts)
                for v in lint_function(name, base, insts):
                    out.append((name,) + v)
    return out, nk, ni


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    path = args[0] if args and os.path.exists(args[0]) else os.path.join(root, "nefes_amd", "libnefes_hip.so")
    flt = args[-1] if args and not os.path.exists(args[-1]) else ""
    viol, nk, ni = lint(path, flt)
    # the probes that exist to violate a rule (csrc/probe.hip store_hazard_kernel<0>, csrc/hazard_probe.hip): counted, not listed
    probes = [v for v in viol if "hazard_probe_kernel<" in v[0] or ("store_hazard_kernel<0>" in v[0] and v[2] == "C6")]
    viol = [v for v in viol if v not in probes]
    by = collections.Counter((v[0][:100], v[2]) for v in viol)
    for (k, rule), n in sorted(by.items()):
        ex = next(v for v in viol if v[0][:100] == k and v[2] == rule)
        print(f"{rule:4s} x{n:<5d} {k}\n       e.g. {ex[1]:#x}: need {ex[3]}, have {ex[4]}:  {ex[5]}   ->   {ex[6]}")
    print(f"{nk} kernels, {ni} instructions, {len(viol)} violations (+ {len(probes)} in the probe kernels that exist to violate a rule)")
    return 1 if viol else 0


if __name__ == "__main__":
    sys.exit(main())
