#!/usr/bin/env python3
"""ISA-level audit of the index-independent kernels: no branch and no memory address may depend on a secret.

The reference promises "no secret-dependent branches or addresses" (/root/reference README.md:92-97) and keeps the promise
with constant_time_lookup (src/include/constant_time.h:134-183, used at src/goldilocks.c:437-442, 864).  This library's
default ("index-independent") kernels keep it by construction -- a Montgomery ladder of selects, LDS combs gathered with
ds_bpermute_b32, table scans of v_cndmask -- but a compiler is free to turn a select chain into a branch or to hoist a
scalar-derived address.  This tool checks what the compiler actually emitted.

Method: every translation unit with such kernels is compiled with `hipcc --offload-device-only -S` (gfx950; cross-compiles
without a GPU) and each audited kernel's ISA goes through a flow-sensitive TAINT ANALYSIS over its control-flow graph:

  * sources: the data behind the kernel arguments listed as secret (scalars, private keys): a load whose address derives
    from such an argument pointer yields tainted registers.  The POINTER is public, the data is not.  A store of tainted
    data through another argument's pointer makes that argument secret as well (results, parked nonces: fixed point).
  * propagation: any instruction with a tainted source (VGPR, SGPR, VCC, SCC, EXEC) taints its destinations; a write
    from clean sources under a clean EXEC cleans its destination; LDS is one cell (tainted once tainted data is written
    to it), scratch is tracked per constant offset; SGPR spill lanes (v_writelane / v_readlane) are tracked per lane.
  * violations:  a conditional branch on a tainted SCC / VCC / EXEC;  a memory instruction (global, scratch, LDS, scalar)
    whose ADDRESS operand is tainted, or which executes under a tainted EXEC;  s_load / s_buffer_load from a tainted
    address;  v_readlane / v_writelane with a tainted lane select.
  * allowed by design: ds_bpermute_b32 / ds_permute_b32, DPP and v_readlane with a tainted DATA operand -- they move
    registers between lanes and touch no memory address (ds_bpermute's "address" is a lane number; this is the wavefront
    shuffle gather the design prescribes).

A write under partial but clean EXEC is taken to replace the register (the lanes left out hold a dead value in
compiler-generated code).  Calls (s_swappc_b64) are not followed: a kernel with one fails the audit.

    python tools/isa_audit.py                 audit everything in AUDIT, print a table, exit 1 on any violation
    python tools/isa_audit.py --kernel k_x448 --verbose
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libgoldilocks_amd", "csrc")
ISA_DIR = os.path.join(ROOT, "build", "isa")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_loops  # noqa: E402

# kernel -> (translation unit, indices of the arguments that point at secret data, {allowed exceptions})
# An exception is (kind, argument index the address derives from) -> why it is allowed.
AUDIT = {
    # variable base, table-free Montgomery ladder (montgomery.hpp)
    "k_point_scalarmul_ct": ("kernels_varbase_ct.hip", [2], {}),
    "k_direct_scalarmul_ct": ("kernels_varbase_ct.hip", [3], {}),
    "k_point_dual_scalarmul_ct": ("kernels_varbase_ct.hip", [3, 4], {}),
    "k_double_scalarmul_ct": ("kernels_varbase_ct.hip", [2, 4], {}),
    # fixed base: combs in LDS gathered with ds_bpermute_b32
    "k_precomputed_scalarmul": ("kernels_fixed.hip", [2], {}),
    "k_base_scalarmul_ct": ("kernels_fixed.hip", [2], {}),
    "k_ed448_derive_public_key_ct": ("kernels_fixed_ct.hip", [1], {}),
    "k_ed448_sign_ct": ("kernels_fixed_ct.hip", [1], {}),
    "k_x448_derive_ct": ("kernels_fixed_ct.hip", [1], {}),
    # X448 with a peer's point: a ladder of selects
    "k_x448": ("kernels_fixed.hip", [3], {
        ("address", 5): "base == NULL is X448 key generation with GOLDILOCKS_AMD_TABLES_FAST (opt-in: digit-addressed window "
                        "table of the base point); the default mode launches k_x448_derive_ct for it",
    }),
    # one operation per wavefront (wave_coop.hpp): window tables in LDS, every entry read for every digit
    "k_point_scalarmul_wave": ("kernels_wave.hip", [2], {}),
    "k_double_scalarmul_wave": ("kernels_wave.hip", [2, 4], {
        ("address", 6): "b1 == NULL is goldilocks_448_base_double_scalarmul_non_secret (src/goldilocks.c:1260-1330): "
                        "public scalars by contract, the base point's half reads its window table by the digit",
    }),
    "k_point_dual_scalarmul_wave": ("kernels_wave.hip", [3, 4], {}),
    "k_direct_scalarmul_wave": ("kernels_wave.hip", [3], {}),
    "k_precomputed_scalarmul_wave": ("kernels_wave.hip", [2], {}),
    "k_derive_wave": ("kernels_wave.hip", [1], {}),
    "k_ed448_sign_wave": ("kernels_wave.hip", [1], {}),
    "k_x448_wave": ("kernels_wave.hip", [3], {}),
    # the reference's scalar API (src/scalar.c): secret scalars are its everyday arguments (a: scalars or their bytes, b)
    "k_scalar_op": ("kernels_misc.hip", [2, 3], {}),
    "k_ed448_expand_secret": ("kernels_misc.hip", [1], {}),
}

# Negative controls: the digit-addressed ("fast", opt-in) twins of two audited kernels.  The audit MUST flag them --
# their table addresses are the secret digits -- or the tool has gone blind (tests/test_isa_audit.py).
CONTROLS = {
    "k_base_scalarmul": ("kernels_fixed.hip", [2], {}),           # the base point's 16-bit window table, read by the digit
    "k_ed448_sign": ("kernels_fixed.hip", [1], {}),               # signing through the same table
}

# ------------------------------------------------------------------------------------------------ compile


def compile_isa(tu, force=False):
    """hipcc -S of one translation unit (cached in build/isa/, redone when any source is newer)"""
    os.makedirs(ISA_DIR, exist_ok=True)
    out = os.path.join(ISA_DIR, tu.replace(".hip", ".s"))
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    if not force and os.path.exists(out) and all(os.path.getmtime(s) <= os.path.getmtime(out) for s in srcs):
        return out
    # the shipped library's own flag list (__graft_entry__.lib_flags): the audited ISA is the ISA of the .so
    sys.path.insert(0, ROOT)
    from __graft_entry__ import lib_flags
    cmd = [HIPCC] + lib_flags() + ["--offload-device-only", "-S", "-o", out + ".tmp", os.path.join(CSRC, tu)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    os.replace(out + ".tmp", out)
    return out


def kernel_arg_offsets(text, kernel):
    """byte offsets of the explicit arguments of `kernel`, in order, from the code object metadata"""
    m = re.search(r"amdhsa\.kernels:(.*)", text, re.S)
    offs = []
    for block in re.split(r"\n  - ", m.group(1)):
        if not re.search(r"\.name:\s+%s\b" % re.escape(kernel), block):
            continue
        args = block.split(".args:")[1].split("\n    .")[0]
        for a in re.split(r"\n      - ", args):
            kind = re.search(r"\.value_kind:\s+(\S+)", a)
            off = re.search(r"\.offset:\s+(\d+)", a)
            size = re.search(r"\.size:\s+(\d+)", a)
            if kind and off and not kind.group(1).startswith("hidden"):
                offs.append((int(off.group(1)), int(size.group(1)), kind.group(1) == "global_buffer"))
        return offs
    raise KeyError(kernel)


def kernarg_sgpr(text, kernel):
    """index of the first of the two user SGPRs that hold the kernel-argument segment's address: they follow the private
    segment buffer (4 registers), the dispatch packet's pointer (2) and the queue's (2) where the kernel descriptor asks for
    those (a kernel that reads its grid size from the dispatch packet has the arguments behind s[2:3], not s[0:1])"""
    m = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(kernel), text, re.S)
    if not m:
        return 0
    flag = lambda name: int((re.search(r"\.amdhsa_user_sgpr_%s\s+(\d+)" % name, m.group(1)) or [0, 0])[1])
    return 4 * flag("private_segment_buffer") + 2 * flag("dispatch_ptr") + 2 * flag("queue_ptr")


# ------------------------------------------------------------------------------------------------ operands

REG = re.compile(r"^(-|\|)?(v|s|a|ttmp)(\d+)\|?$")
RANGE = re.compile(r"^(-|\|)?(v|s|a|ttmp)\[(\d+):(\d+)\]\|?$")
SPECIAL = {"vcc": ["vcc"], "vcc_lo": ["vcc"], "vcc_hi": ["vcc"], "exec": ["exec"], "exec_lo": ["exec"], "exec_hi": ["exec"],
           "scc": ["scc"], "m0": ["m0"]}


def regs_of(op):
    """registers named by one operand ('' for immediates, labels, off, ...)"""
    op = op.strip()
    m = REG.match(op)
    if m:
        return ["%s%d" % (m.group(2), int(m.group(3)))]
    m = RANGE.match(op)
    if m:
        return ["%s%d" % (m.group(2), i) for i in range(int(m.group(3)), int(m.group(4)) + 1)]
    return SPECIAL.get(op, [])


def split_operands(rest):
    """'v[0:1], s[2:3], v4, v5 offset:16 glc' -> (['v[0:1]', 's[2:3]', 'v4', 'v5'], ['offset:16', 'glc'])"""
    rest = rest.split(";")[0].strip()
    ops, mods, depth, cur = [], [], 0, ""
    for ch in rest:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    if ops:
        parts = ops[-1].split()
        if parts:
            ops[-1] = parts[0]
            mods = parts[1:]
    # a modifier can also trail an operand that is not the last ("v1 row_shr:1" only appears last in practice)
    return ops, mods


class Ins:
    __slots__ = ("line", "text", "mn", "ops", "mods", "idx")

    def __init__(self, line, text):
        self.line, self.text = line, text.strip()
        self.idx = None          # (index SGPR, "SRC0,DST,...") between s_set_gpr_idx_on and s_set_gpr_idx_off
        parts = self.text.split(None, 1)
        self.mn = parts[0]
        self.ops, self.mods = split_operands(parts[1]) if len(parts) > 1 else ([], [])


# ------------------------------------------------------------------------------------------------ semantics

TWO_DST = re.compile(r"^(v_mad_[ui]64_[ui]32|v_(add|sub|subrev)_co_u32_e64|v_(addc|subb|subbrev)_co_u32_e64|v_div_scale_\w+)$")
NO_SCC = {"s_mov_b32", "s_mov_b64", "s_movk_i32", "s_cselect_b32", "s_cselect_b64", "s_mul_i32", "s_mul_hi_u32", "s_mul_hi_i32",
          "s_getpc_b64", "s_setpc_b64", "s_swappc_b64", "s_brev_b32", "s_brev_b64", "s_cmov_b32", "s_cmov_b64", "s_cmovk_i32",
          "s_sext_i32_i8", "s_sext_i32_i16", "s_bitset0_b32", "s_bitset1_b32", "s_pack_ll_b32_b16"}
READS_SCC = {"s_cselect_b32", "s_cselect_b64", "s_addc_u32", "s_subb_u32", "s_cmov_b32", "s_cmov_b64", "s_cmovk_i32"}
NOPS = {"s_set_gpr_idx_on", "s_set_gpr_idx_off", "s_set_gpr_idx_mode", "s_nop", "s_waitcnt", "s_barrier", "s_sleep", "s_setprio", "s_sethalt", "s_endpgm", "s_code_end", "s_setreg_b32",
        "s_setreg_imm32_b32", "s_waitcnt_vscnt", "s_icache_inv", "s_dcache_wb", "s_inst_prefetch", "s_clause", "s_trap",
        "s_sendmsg", "s_ttracedata", "s_waitcnt_depctr", "buffer_wbl2", "buffer_inv", "s_dcache_inv"}
LANE_MOVES = {"ds_bpermute_b32", "ds_permute_b32", "ds_swizzle_b32"}


class Violation:
    def __init__(self, kind, ins, why, prov=()):
        self.kind, self.ins, self.why, self.prov = kind, ins, why, tuple(sorted(prov))

    def __str__(self):
        return "%-8s line %d: %s   [%s]" % (self.kind, self.ins.line, self.ins.text, self.why)


class State:
    """register -> (tainted, provenance = frozenset of argument indices the value derives from as a pointer)"""

    def __init__(self):
        self.t = {}

    def copy(self):
        s = State()
        s.t = dict(self.t)
        return s

    def get(self, r):
        return self.t.get(r, (False, frozenset()))

    def set(self, r, v):
        if v == (False, frozenset()):
            self.t.pop(r, None)
        else:
            self.t[r] = v

    def join(self, other):
        """self |= other; True if self changed"""
        changed = False
        for r, (t, p) in other.t.items():
            t0, p0 = self.get(r)
            n = (t0 or t, p0 | p)
            if n != (t0, p0):
                self.t[r] = n
                changed = True
        return changed


def combine(vals):
    t, p = False, frozenset()
    for a, b in vals:
        t, p = t or a, p | b
    return t, p


class Audit:
    def __init__(self, kernel, body, first_line, arg_offsets, secret_args, verbose=False, kernarg_at=0):
        self.kernarg_at = kernarg_at
        self.kernel, self.verbose = kernel, verbose
        self.arg_offsets = arg_offsets
        self.secret = set(secret_args)        # grows: arguments through which tainted data was stored
        self.lds_tainted = False
        self.scratch_any = False               # tainted data was stored to scratch at a register address
        self.ins, self.labels = [], {}
        for k, l in enumerate(body):
            m = re.match(r"^(\.L[A-Za-z0-9_$.]+):", l)
            if m:
                self.labels[m.group(1)] = len(self.ins)
            elif "implicit-def: $" in l:
                # the compiler's own note that these registers hold no defined value here (a value assigned on every
                # FEASIBLE path below, e.g. in two correlated branches): whatever they held before is dead
                regs = []
                for name in re.findall(r"\$((?:[vsa]gpr\d+_?)+|vcc|exec)", l):
                    for m in re.finditer(r"([vsa])gpr(\d+)", name):
                        regs.append("%s%s" % (m.group(1), m.group(2)))
                    if name in ("vcc", "exec"):
                        regs.append(name)
                ins = Ins(first_line + k, "__kill " + ", ".join(regs))
                self.ins.append(ins)
            elif isa_loops.is_instr(l) and not l.strip().startswith(";;#"):
                self.ins.append(Ins(first_line + k, l))
        # VGPR indexing mode (s_set_gpr_idx_on sN, gpr_idx(SRC0) ... s_set_gpr_idx_off): the VALU instructions in between
        # read / write v[operand + sN]
        mode = None
        for ins in self.ins:
            if ins.mn == "s_set_gpr_idx_on":
                mode = (ins.ops[0], ins.ops[1] if len(ins.ops) > 1 else "")
            elif ins.mn == "s_set_gpr_idx_off":
                mode = None
            elif mode and ins.mn.startswith("v_"):
                ins.idx = mode
        self.unhandled = set()
        self.exec_predication = True     # a VALU / memory write under a tainted EXEC taints its destination
        self.build_cfg()

    # -- control flow
    def build_cfg(self):
        n = len(self.ins)
        self.succ = [[] for _ in range(n)]
        for i, ins in enumerate(self.ins):
            mn = ins.mn
            if mn == "s_endpgm":
                continue
            if mn == "s_branch":
                self.succ[i] = [self.labels[ins.ops[0]]]
            elif mn.startswith("s_cbranch"):
                self.succ[i] = [self.labels[ins.ops[0]]] + ([i + 1] if i + 1 < n else [])
            elif mn == "s_setpc_b64":
                # a long branch: s_getpc_b64 / s_add_u32 (.LBBx_y-.Lpost_getpcN) / s_addc_u32 / s_setpc_b64
                target = None
                for j in range(i - 1, max(i - 6, -1), -1):
                    m = re.search(r"\((\.L[A-Za-z0-9_$.]+)-\.Lpost_getpc\d+\)", self.ins[j].text)
                    if m:
                        target = m.group(1)
                        break
                if target is None:
                    raise ValueError("s_setpc_b64 that is not a long branch at line %d" % ins.line)
                self.succ[i] = [self.labels[target]]
            elif i + 1 < n:
                self.succ[i] = [i + 1]

    # -- one instruction
    def argument_of(self, byte):
        for k, (off, size, is_pointer) in enumerate(self.arg_offsets):
            if off <= byte < off + size:
                return k if is_pointer else None     # only a pointer argument gives its value a provenance
        return None

    def src_val(self, st, ops):
        return combine(st.get(r) for op in ops for r in regs_of(op))

    def addr_of(self, ops):
        """address operands of a memory instruction as a flat register list"""
        return [r for op in ops for r in regs_of(op)]

    def step(self, st, ins, out):
        """transfer function; appends Violations to `out` when out is not None"""
        mn, ops = ins.mn, ins.ops
        exec_t = st.get("exec")[0]

        def violate(kind, why, prov=()):
            if out is not None:
                out.append(Violation(kind, ins, why, prov))

        def write(regs, val, predicated):
            t, p = val
            if predicated and exec_t and self.exec_predication:
                t = True
            for r in regs:
                st.set(r, (t, p))

        if mn in NOPS or mn.startswith("s_waitcnt"):
            return
        if mn == "__kill":
            for op in ops:
                for r in regs_of(op):
                    st.set(r, (False, frozenset()))
            return
        # ---- branches
        if mn.startswith("s_cbranch"):
            cond = {"scc": "scc", "vcc": "vcc", "exe": "exec"}[mn.split("_")[2][:3]]
            if st.get(cond)[0]:
                violate("branch", "condition %s depends on a secret" % cond)
            return
        if mn in ("s_branch", "s_setpc_b64", "s_getpc_b64"):
            if mn == "s_getpc_b64":
                write(regs_of(ops[0]), (False, frozenset()), False)
            return
        if mn == "s_swappc_b64":
            violate("call", "calls are not followed")
            return
        # ---- scalar memory
        if mn.startswith("s_load_") or mn.startswith("s_buffer_load") or mn.startswith("s_scratch_load"):
            dst, base = regs_of(ops[0]), regs_of(ops[1])
            off_regs = regs_of(ops[2]) if len(ops) > 2 else []
            at, ap = combine(st.get(r) for r in base + off_regs)
            if at:
                violate("address", "scalar load from a secret-dependent address", ap)
            # the kernarg segment: its two user SGPRs at entry (kernarg_sgpr), or a copy of them (a kernel with many arguments
            # moves the pointer aside before the registers are reused: the copy carries the provenance "kernarg")
            if not off_regs and ap == frozenset(["kernarg"]):
                imm = int(ops[2], 0) if len(ops) > 2 else 0
                for k, r in enumerate(dst):
                    a = self.argument_of(imm + 4 * k)
                    st.set(r, (False, frozenset() if a is None else frozenset([a])))
            else:
                t = at or bool(ap & self.secret)
                write(dst, (t, frozenset()), False)
            return
        # ---- vector / LDS / scratch memory
        if mn.startswith(("global_load", "flat_load", "global_atomic", "flat_atomic", "buffer_load", "buffer_atomic")):
            dst = regs_of(ops[0])
            addr = self.addr_of(ops[1:])
            at, ap = combine(st.get(r) for r in addr)
            if at:
                violate("address", "load from a secret-dependent address", ap)
            if exec_t:
                violate("exec", "memory access under a secret-dependent EXEC", ap)
            if "atomic" in mn:
                violate("address", "atomic in an audited kernel (not modelled)", ap)
            write(dst, (at or bool(ap & self.secret), frozenset()), True)
            return
        if mn.startswith(("global_store", "flat_store", "buffer_store")):
            addr = self.addr_of([ops[0]] + ops[2:])
            data = self.src_val(st, [ops[1]])
            at, ap = combine(st.get(r) for r in addr)
            if at:
                violate("address", "store to a secret-dependent address", ap)
            if exec_t:
                violate("exec", "memory access under a secret-dependent EXEC", ap)
            if data[0] and not ap <= self.secret:
                self.secret |= ap
                self.changed = True
            return
        if mn.startswith(("scratch_load", "scratch_store")):
            # Spill slots are registers by another name: a slot at a constant offset is tracked in the state like one
            # (strong update, flow-sensitive -- the allocator reuses a slot for unrelated values); a scratch access with
            # a register address falls back to one cell for all of scratch.
            load = mn.startswith("scratch_load")
            width = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4, "short": 1, "byte": 1, "ubyte": 1, "sbyte": 1,
                     "ushort": 1, "sshort": 1}[mn.split("_")[-1]]
            addr_regs = self.addr_of(ops[1:]) if load else self.addr_of([ops[0]] + ops[2:])
            at, _ = combine(st.get(r) for r in addr_regs)
            if at:
                violate("address", "scratch access at a secret-dependent address")
            if exec_t:
                violate("exec", "scratch access under a secret-dependent EXEC")
            base = self.const_offset(ins)
            if load:
                dst = regs_of(ops[0])
                for k, r in enumerate(dst):
                    if addr_regs:
                        v = (self.scratch_any or at, frozenset())
                    else:
                        v = st.get("scratch@%d" % (base + 4 * min(k, width - 1)))
                        v = (v[0] or self.scratch_any, v[1])
                    write([r], v, True)
            else:
                data = [st.get(r) for r in regs_of(ops[1])]
                if addr_regs:
                    if any(t for t, _ in data) and not self.scratch_any:
                        self.scratch_any = True
                        self.changed = True
                else:
                    for k, v in enumerate(data):
                        st.set("scratch@%d" % (base + 4 * k), (v[0] or (exec_t and self.exec_predication), v[1]))
            return
        if mn in LANE_MOVES:
            # lane number (or swizzle pattern) in the "address" operand: a register move between lanes, no memory address
            dst = regs_of(ops[0])
            write(dst, self.src_val(st, ops[1:]), True)
            return
        if mn.startswith(("ds_read", "ds_load")):
            dst = regs_of(ops[0])
            at, _ = combine(st.get(r) for r in self.addr_of(ops[1:]))
            if at:
                violate("address", "LDS read from a secret-dependent address")
            if exec_t:
                violate("exec", "LDS access under a secret-dependent EXEC")
            write(dst, (self.lds_tainted or at, frozenset()), True)
            return
        if mn.startswith(("ds_write", "ds_store")):
            at, _ = st.get(regs_of(ops[0])[0]) if regs_of(ops[0]) else (False, None)
            if at:
                violate("address", "LDS write to a secret-dependent address")
            if exec_t:
                violate("exec", "LDS access under a secret-dependent EXEC")
            if self.src_val(st, ops[1:])[0] and not self.lds_tainted:
                self.lds_tainted = True
                self.changed = True
            return
        if mn.startswith(("ds_", "image_", "tbuffer_", "exp")):
            self.unhandled.add(mn)
            violate("unknown", "instruction class not modelled")
            return
        # ---- lane spills and lane reads
        if mn == "v_writelane_b32":
            lane = ops[2]
            if regs_of(lane) and st.get(regs_of(lane)[0])[0]:
                violate("address", "v_writelane_b32 with a secret-dependent lane select")
            v = regs_of(ops[0])[0]
            key = "%s.lane%s" % (v, lane) if not regs_of(lane) else v
            st.set(key, self.src_val(st, [ops[1]]))
            return
        if mn == "v_readlane_b32":
            lane = ops[2]
            if regs_of(lane) and st.get(regs_of(lane)[0])[0]:
                violate("address", "v_readlane_b32 with a secret-dependent lane select")
            v = regs_of(ops[1])[0]
            key = "%s.lane%s" % (v, lane)
            val = st.get(key) if (not regs_of(lane) and key in st.t) else st.get(v)
            if not regs_of(lane) and key not in st.t and any(k.startswith(v + ".lane") for k in st.t):
                val = (False, frozenset())     # a spill lane that was never written on this path
            write(regs_of(ops[0]), val, False)
            return
        # ---- EXEC manipulation
        if re.match(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor|andn1|orn1)_saveexec_b64$", mn):
            val = combine([self.src_val(st, [ops[1]]), st.get("exec")])
            write(regs_of(ops[0]), st.get("exec"), False)
            st.set("exec", val)
            st.set("scc", val)
            return
        if mn.startswith("v_cmpx"):
            val = combine([self.src_val(st, ops), st.get("exec")])
            st.set("exec", val)
            if ops and ops[0] in ("vcc",) or (ops and regs_of(ops[0]) and regs_of(ops[0])[0].startswith("s")):
                write(regs_of(ops[0]), val, False)
            return
        # ---- ALU, generic
        if mn.startswith("v_") or mn.startswith("s_"):
            if not ops:
                self.unhandled.add(mn)
                return
            ndst = 2 if TWO_DST.match(mn) else 1
            if re.match(r"^v_(add|sub|subrev|addc|subb|subbrev)_co_u32_e32$", mn):
                ndst = 2                                               # vdst, vcc, a, b[, vcc]
            if mn.startswith("s_cmp") or mn.startswith("s_bitcmp"):
                st.set("scc", self.src_val(st, ops))
                return
            dst_ops, src_ops = ops[:ndst], ops[ndst:]
            srcs = [self.src_val(st, src_ops)]
            if ins.idx:
                # relative VGPR addressing: which register is read is decided by an SGPR.  A secret index is reported
                # (a register file "address"); with a public index the operand is any of the registers above its base
                idx_reg, which = ins.idx
                if regs_of(idx_reg) and st.get(regs_of(idx_reg)[0])[0]:
                    violate("index", "VGPR index mode with a secret-dependent index")
                for k, op in enumerate(src_ops):
                    if "SRC%d" % k in which and regs_of(op) and regs_of(op)[0].startswith("v"):
                        base = int(regs_of(op)[0][1:])
                        srcs.append(combine(st.get("v%d" % r) for r in range(base, min(base + 64, 256))))
                if "DST" in which:
                    self.unhandled.add("s_set_gpr_idx_on(DST)")
            if mn in READS_SCC:
                srcs.append(st.get("scc"))
            partial = ("UNUSED_PRESERVE" in ins.text                              # SDWA writing part of the register
                       or (not mn.startswith("v_pk") and re.search(r"op_sel:\[[01,]*1\]", ins.text) is not None))   # high half only
            if mn in ("v_mac_f32_e32", "v_fmac_f32_e32", "v_fmac_f64_e32") or partial:
                srcs.append(self.src_val(st, dst_ops[:1]))               # the destination is (partly) kept
            if mn.endswith("_dpp"):
                # DPP: a lane keeps its old destination when it is masked off (row_mask / bank_mask) or when its source
                # lane does not exist (a shift without bound_ctrl); rotations and quad_perm write every lane
                mods = " ".join(ins.mods)
                masked = re.search(r"row_mask:0x[0-9a-e]\b|bank_mask:0x[0-9a-e]\b", mods) is not None
                shifts = re.search(r"\b(row_shl|row_shr|wave_shl|wave_shr|row_bcast)", mods) is not None
                if masked or (shifts and "bound_ctrl" not in mods):
                    srcs.append(self.src_val(st, dst_ops[:1]))
            val = combine(srcs)
            is_v = mn.startswith("v_")
            if mn == "v_readfirstlane_b32":     # an SGPR result: not predicated, but WHICH lane is first depends on EXEC
                is_v = False
                if exec_t:
                    val = (True, val[1])
            # pointers stay pointers only through address arithmetic; anything else drops the provenance
            if not re.match(r"^(v_mov_b32|v_mov_b64|s_mov_b32|s_mov_b64|v_add_co_u32|v_addc_co_u32|v_add_u32|v_lshl_add_u64|"
                            r"v_mad_u64_u32|v_mad_i64_i32|s_add_u32|s_addc_u32|s_add_i32|v_add3_u32|v_lshl_add_u32|v_add_lshl_u32|"
                            r"v_readfirstlane_b32|v_cndmask_b32|s_cselect_b32|s_cselect_b64|v_or_b32|v_or3_b32|v_lshl_or_b32|"
                            r"v_and_or_b32|v_sub_co_u32|v_subb_co_u32|v_sub_u32|s_sub_u32|s_subb_u32|v_mad_u32_u24|v_accvgpr_\w+)(_e32|_e64|_dpp)?$", mn):
                val = (val[0], frozenset())
            for op in dst_ops:
                write(regs_of(op), val, is_v)
            if mn.startswith("s_") and mn not in NO_SCC:
                st.set("scc", (val[0], frozenset()))
            return
        self.unhandled.add(mn)
        violate("unknown", "instruction not modelled")

    @staticmethod
    def const_offset(ins):
        for m in ins.mods:
            if m.startswith("offset:"):
                return int(m.split(":")[1], 0)
        return 0

    # -- fixed point over the CFG, repeated until the flow-insensitive parts (secret arguments, LDS, scratch) settle
    def run(self):
        n = len(self.ins)
        while True:
            self.changed = False
            entry = [None] * n
            entry[0] = State()
            entry[0].set("s%d" % self.kernarg_at, (False, frozenset(["kernarg"])))
            entry[0].set("s%d" % (self.kernarg_at + 1), (False, frozenset(["kernarg"])))
            work = [0]
            while work:
                i = work.pop()
                st = entry[i].copy()
                self.step(st, self.ins[i], None)
                for j in self.succ[i]:
                    if entry[j] is None:
                        entry[j] = st.copy()
                        work.append(j)
                    elif entry[j].join(st):
                        work.append(j)
            if not self.changed:
                break
        out = []
        for i in range(n):
            if entry[i] is not None:
                self.step(entry[i].copy(), self.ins[i], out)
        self.reached = sum(1 for e in entry if e is not None)
        self.tainted_loads = sum(1 for i in range(n) if entry[i] is not None and self.ins[i].mn.startswith(("global_load", "ds_read"))
                                 and self._dst_tainted(entry[i], self.ins[i]))
        return out

    def _dst_tainted(self, st, ins):
        s = st.copy()
        self.step(s, ins, None)
        return any(s.get(r)[0] for r in regs_of(ins.ops[0]))


def audit_kernel(kernel, verbose=False, force=False):
    tu, secret_args, exceptions = AUDIT[kernel] if kernel in AUDIT else CONTROLS[kernel]
    path = compile_isa(tu, force)
    text = open(path).read()
    lines = text.split("\n")
    funcs = isa_loops.functions(text)
    if kernel not in funcs:
        raise KeyError("%s not in %s" % (kernel, tu))
    first = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":")) + 2
    a = Audit(kernel, funcs[kernel], first, kernel_arg_offsets(text, kernel), secret_args, verbose, kernarg_sgpr(text, kernel))
    violations = a.run()
    allowed, bad = [], []
    for v in violations:
        why = None
        for (kind, arg), reason in exceptions.items():
            if v.kind == kind and arg in v.prov:
                why = reason
        (allowed if why else bad).append(v)
    stats = {
        "instructions": len(a.ins), "reached": a.reached, "secret_args": sorted(a.secret), "lds_tainted": a.lds_tainted,
        "tainted_loads": a.tainted_loads,
        "branches": sum(1 for i in a.ins if i.mn.startswith("s_cbranch")),
        "memory_instructions": sum(1 for i in a.ins if i.mn.startswith(("global_", "ds_", "scratch_", "s_load", "flat_", "buffer_"))),
        "selects": sum(1 for i in a.ins if i.mn.startswith("v_cndmask")),
        "lane_moves": sum(1 for i in a.ins if i.mn in LANE_MOVES),
        "unhandled": sorted(a.unhandled),
    }
    return bad, allowed, stats


def main(argv):
    verbose = "--verbose" in argv
    force = "--force" in argv
    kernels = [argv[argv.index("--kernel") + 1]] if "--kernel" in argv else list(AUDIT)
    # compile the translation units in parallel first
    tus = sorted({(AUDIT.get(k) or CONTROLS[k])[0] for k in kernels})
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(4, len(tus))) as ex:
        list(ex.map(lambda t: compile_isa(t, force), tus))
    failed = 0
    print("%-30s %8s %8s %7s %8s %6s %5s  %s" % ("kernel", "instr", "branches", "memory", "selects", "lane", "bad", "secret args (after the fixed point)"))
    for k in kernels:
        bad, allowed, s = audit_kernel(k, verbose)
        print("%-30s %8d %8d %7d %8d %6d %5d  %s%s" % (k, s["instructions"], s["branches"], s["memory_instructions"], s["selects"],
                                                      s["lane_moves"], len(bad), s["secret_args"],
                                                      "  (+%d allowed)" % len(allowed) if allowed else ""))
        if s["tainted_loads"] == 0:
            print("    !! no load returned secret data: the sources are wrong")
            failed += 1
        for v in bad[:40 if verbose else 8]:
            print("    " + str(v))
        if verbose:
            for v in allowed[:10]:
                print("    allowed: " + str(v))
        failed += len(bad)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))



def explain(kernel, line, reg=None, depth=40, exec_predication=True):
    """development aid: how did `reg` (default: the first tainted operand) of the instruction at source line `line`
    get tainted?  Walks backwards: the instruction that wrote the register, then that instruction's tainted source, ..."""
    tu, secret_args, _ = AUDIT[kernel] if kernel in AUDIT else CONTROLS[kernel]
    text = open(compile_isa(tu)).read()
    lines = text.split("\n")
    first = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":")) + 2
    a = Audit(kernel, isa_loops.functions(text)[kernel], first, kernel_arg_offsets(text, kernel), secret_args, False, kernarg_sgpr(text, kernel))
    a.exec_predication = exec_predication
    a.run()
    n = len(a.ins)
    entry = [None] * n
    entry[0] = State()
    entry[0].set("s%d" % a.kernarg_at, (False, frozenset(["kernarg"])))
    entry[0].set("s%d" % (a.kernarg_at + 1), (False, frozenset(["kernarg"])))
    work = [0]
    while work:
        i = work.pop()
        st = entry[i].copy()
        a.step(st, a.ins[i], None)
        for j in a.succ[i]:
            if entry[j] is None:
                entry[j] = st.copy()
                work.append(j)
            elif entry[j].join(st):
                work.append(j)
    pred = [[] for _ in range(n)]
    for i in range(n):
        for j in a.succ[i]:
            pred[j].append(i)
    at = next(i for i, x in enumerate(a.ins) if x.line == line)
    if reg is None:
        reg = next(r for op in a.ins[at].ops for r in regs_of(op) if entry[at].get(r)[0])
    seen = set()
    for _ in range(depth):
        # walk back to the instruction whose execution makes `reg` tainted
        i = at
        while True:
            ps = [j for j in pred[i] if entry[j] is not None and (j, reg) not in seen]
            writer = None
            for j in ps:
                st = entry[j].copy()
                a.step(st, a.ins[j], None)
                if st.get(reg)[0]:
                    seen.add((j, reg))
                    if not entry[j].get(reg)[0] or reg in [r for op in a.ins[j].ops[:2] for r in regs_of(op)]:
                        writer = j
                    i = j
                    break
            else:
                print("   (no predecessor taints %s)" % reg)
                return
            if writer is not None:
                break
        ins = a.ins[writer]
        # causal sources: tainted operands (or EXEC / SCC / LDS) whose cleaning would clean `reg`
        tainted = []
        for r in [x for op in ins.ops for x in regs_of(op)] + ["exec", "scc", "vcc"]:
            if not entry[writer].get(r)[0] or r in tainted:
                continue
            st = State()
            for q, (t, p) in entry[writer].t.items():
                st.set(q, (q == r, p))
            a.step(st, ins, None)
            if st.get(reg)[0]:
                tainted.append(r)
        print("%7d  %-70s %s <- %s%s" % (ins.line, ins.text[:70], reg, tainted, "  (EXEC tainted)" if entry[writer].get("exec")[0] else ""))
        nxt = [r for r in tainted if r != reg] or tainted
        if not nxt:
            if ins.mn.startswith(("global_load", "ds_read", "scratch_load", "s_load")):
                print("         a load of secret data: the source")
            return
        at, reg = writer, nxt[0]
