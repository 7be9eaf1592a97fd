#!/usr/bin/env python3
"""Static check of the hand-counted LDS protocol of libmau_hip's kernels, on the ISA the compiler actually emitted.

The convolution / weight-gradient kernels issue their operand-fragment reads from inline asm (``ds_read_b128``,
``ds_read_b64_tr_b16`` pairs) and retire them with hand-counted ``s_waitcnt lgkmcnt(N)``: the compiler does not know these are LDS
reads, so nothing stops it from placing a register copy of a fragment half (``v_mov`` after a ``__builtin_shufflevector`` of the two
transposed 8-byte reads) or a consumer BEFORE the wait -- an allocation-dependent bug class (conv3x3_first.hip ``tr_frag``,
conv3x3_wgrad16.hip / conv3x3_wgrad_bf16.hip ``tr_read``), correct today only because the allocator happens to coalesce the halves.

This script pins that: it disassembles the gfx950 code object of every translation unit and walks each kernel in program order with
the hardware's own model -- LGKM operations of a wave (DS, SMEM) complete in issue order for DS; ``s_waitcnt lgkmcnt(N)`` = at most N
still outstanding -- and reports every instruction that touches (reads OR writes) a VGPR that an outstanding ``ds_read*`` has not
delivered yet.  SMEM loads share the counter and may return out of order: they are kept in the queue as entries without VGPRs, which
only makes a counted wait look LESS complete than it is (conservative).  Loops are walked once in listing order (a read left
outstanding across a back edge is still outstanding at the loop head of the listing, which is where the next iteration's first wait is
checked against it).

    python scripts/check_lds_waits.py [objects ...]      default: every csrc/*.o of the in-tree build; exit code 1 on a hazard
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "metadata-augmented-unet-for-lst-ndvi_amd", "csrc")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def code_object(obj, tmp):
    """-> path of the gfx950 code object bundled in a host object file ('' if it carries none)"""
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return ""
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"],
                       capture_output=True, text=True)
    return co if (r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0) else ""


def kernel_resources(obj, tmp):
    """[(kernel name, VGPRs, spilled VGPRs, scratch bytes per lane)] from the code object's metadata.  The matrix-core kernels must use
    NO scratch: a spilled register is reloaded by a vector-memory load, which besides its latency sits in the vmcnt that the kernels'
    hand-counted DMA waits count (round 5 found 8-26 spilled registers in five instantiations of the convolution kernel -- loop-invariant
    slot coordinates hoisted out of the item loop -- invisible in the source)."""
    co = code_object(obj, tmp)
    if not co:
        return []
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"\s*\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name":
            if cur.get("name") and "vgpr_count" in cur:
                out.append(cur)
            cur = {"name": v}
        else:
            cur[k] = int(v)
    if cur.get("name") and "vgpr_count" in cur:
        out.append(cur)
    return [(c["name"], c.get("vgpr_count", -1), c.get("vgpr_spill_count", 0), c.get("private_segment_fixed_size", 0)) for c in out]


def disassemble(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return ""
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        return ""
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def merge(q1, q2):
    """Two paths meet: an entry stays outstanding if it is on EITHER path, aligned from the most recent operation backwards (a counted
    wait speaks about recency: ``lgkmcnt(N)`` = everything but the last N has landed)."""
    if q1 is None:
        return [set(e) for e in q2]
    if q2 is None:
        return q1
    n = max(len(q1), len(q2))
    a = [set()] * (n - len(q1)) + q1
    b = [set()] * (n - len(q2)) + q2
    return [x | y for x, y in zip(a, b)]


def check(asm, name):
    """-> (hazards, number of ds reads seen).  Forward branches are followed (the state of the branching path is carried to the
    target; code behind an unconditional branch is entered only with the states that jump there); back edges are not."""
    hazards, reads = [], 0
    func, start, queue, snap = None, 0, [], {}   # queue entries: set of VGPRs an outstanding LGKM op will write (empty for writes / SMEM)
    for line in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
        if m:
            func, start, queue, snap = m.group(2), int(m.group(1), 16), [], {}
            continue
        parts = line.split("//")
        ins = parts[0].strip()
        if not ins or func is None:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", parts[1]) if len(parts) > 1 else None
        addr = int(am.group(1), 16) if am else None
        if addr is not None and addr in snap:
            queue = merge(queue, snap.pop(addr))
        if queue is None:                        # behind an unconditional branch and no forward jump lands here: the target of a back
            queue = []                           # edge (a rotated loop's body) -- walked as entered with nothing outstanding
        op = ins.split()[0]
        if op == "s_endpgm":
            queue = None
            continue
        if op in ("s_branch",) or op.startswith("s_cbranch"):
            tm = re.search(r"<.*\+0x([0-9a-fA-F]+)>", line)
            if tm:
                tgt = start + int(tm.group(1), 16)
                if addr is not None and tgt > addr:
                    snap[tgt] = merge(snap.get(tgt), [set(e) for e in queue])
            if op == "s_branch":
                queue = None
            continue
        if op.startswith("s_waitcnt"):
            m = LGKM.search(ins)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n < len(queue) else queue
            elif re.fullmatch(r"s_waitcnt\s+0(x0+)?", ins):       # (all counters zero, printed raw)
                queue = []
            continue
        touched = vregs(ins.split(None, 1)[1]) if " " in ins else set()
        pending = set().union(*queue) if queue else set()
        if op.startswith("ds_read") or op.startswith("ds_load"):
            ops = ins.split(None, 1)[1]
            dst = vregs(ops.split(",")[0])
            addr_regs = touched - dst
            if addr_regs & pending:
                hazards.append((name, func, ins, sorted(addr_regs & pending)))
            if dst & pending:                                     # two outstanding reads into one register: order of arrival decides
                hazards.append((name, func, ins, sorted(dst & pending)))
            queue.append(dst)
            reads += 1
            continue
        if touched & pending:
            hazards.append((name, func, ins, sorted(touched & pending)))
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_sendmsg") or op.startswith("s_memtime"):
            queue.append(set())
    return hazards, reads


def scratch_in_mfma_region(asm):
    """Scratch (spill) accesses BETWEEN the first and the last MFMA of a kernel's listing -- i.e. inside its multiply loops, where a
    reload's latency stalls the matrix pipes and its vmcnt slot breaks the hand-counted DMA waits.  -> [(kernel, instruction)]"""
    out, func, seen_mfma, pending = [], None, False, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func, seen_mfma, pending = m.group(1), False, []
            continue
        ins = line.split("//")[0].strip()
        if not ins or func is None:
            continue
        op = ins.split()[0]
        if op.startswith("v_mfma"):
            out += pending                       # scratch accesses seen since the previous MFMA lie between two MFMAs
            pending, seen_mfma = [], True
        elif op.startswith("scratch_") and seen_mfma:
            pending.append((func, ins))
    return out


def main(argv):
    objs = argv or sorted(glob.glob(os.path.join(CSRC, "*.o")))
    objs = [o for o in objs if not o.endswith(".asan.o")]
    total_reads, all_h = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            asm = disassemble(o, tmp)
            if not asm:
                continue
            h, r = check(asm, os.path.basename(o))
            total_reads += r
            all_h += h
            print(f"{os.path.basename(o):32s} ds reads {r:6d}  hazards {len(h)}")
    for name, func, ins, regs in all_h[:40]:
        print(f"HAZARD {name} {func[:90]}: `{ins}` touches v{regs} before the LDS read that writes it is waited for")
    print(f"{total_reads} LDS reads checked, {len(all_h)} hazards")
    return 1 if all_h else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
