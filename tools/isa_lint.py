#!/usr/bin/env python
"""tools/isa_lint.py -- compile the two Winograd F(4x4) kernels to gfx950 assembly (no GPU needed) and report, per instantiation, the
things round 4 found to cost time on this chip because loads, stores and scratch reloads share ONE in-order counter (vmcnt):

  * spilled VGPRs (a reload is a VMEM operation: it waits for everything issued before it, and everything after it waits for it);
  * scratch reloads BETWEEN the global stores of an output round (each is a drain of the stores in front of it);
  * `s_waitcnt vmcnt(0)` inside the plane loop (a drain of the weight ring in flight).

  python tools/isa_lint.py            # prints one line per kernel; exit code 1 if a 64-column Cylindrical_Net kernel regresses
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "buffer-x_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CS,
         "-S", "--cuda-device-only"]


def kernels(asm):
    """{mangled name: [instruction lines]} of every kernel in an assembly file"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None and line.startswith("\t") and not line.lstrip().startswith((";", ".")):
            out[cur].append(line.strip())
    return out


def report(name, ins, spills):
    stores = [i for i, l in enumerate(ins) if l.startswith(("global_store", "buffer_store"))]
    between = 0
    if stores:
        # output rounds = runs of stores without an s_barrier in between
        run = [stores[0]]
        for a, b in zip(stores, stores[1:]):
            if any(l.startswith("s_barrier") for l in ins[a:b]):
                between += sum(1 for l in ins[run[0]:run[-1]] if l.startswith("scratch_load"))
                run = [b]
            else:
                run.append(b)
        between += sum(1 for l in ins[run[0]:run[-1]] if l.startswith("scratch_load"))
    mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    drains = 0
    if mf:
        # the plane loop = between the first and the last MFMA of the longest barrier-free run of MFMAs
        runs, cur = [], [mf[0]]
        for a, b in zip(mf, mf[1:]):
            if any(l.startswith("s_barrier") for l in ins[a:b]):
                runs.append(cur)
                cur = [b]
            else:
                cur.append(b)
        runs.append(cur)
        longest = max(runs, key=len)
        drains = sum(1 for l in ins[longest[0]:longest[-1]] if re.match(r"s_waitcnt\s+vmcnt\(0\)", l))
    # round 5 (two slots since round 6): a VALU write to the data registers of a 16-byte store within the next two wait states.  hipcc leaves no wait state there when the
    # store carries an SGPR offset, and gfx950 then stores half-overwritten data (k_wino43.hip::wino43_output): must be 0
    hazard = 0
    for i, l in enumerate(ins[:-1]):
        m = re.match(r"(?:buffer|global)_store_dwordx[34] (?:v\d+, |v\[\d+:\d+\], )?v\[(\d+):(\d+)\]", l) or re.match(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\]", l)
        if not m:
            continue
        a, b = int(m.group(1)), int(m.group(2))
        # LLVM's rule for the gfx940 family: TWO wait states between a > 64-bit store and a VALU write of its data registers.  Walk the
        # issue slots behind the store: an `s_nop N` is N + 1 states, any other instruction one; a VALU write inside the first two is a hazard.
        states, j = 0, i + 1
        while j < len(ins) and states < 2:
            nxt = ins[j]
            nop = re.match(r"s_nop (\d+)", nxt)
            if nop:
                states += int(nop.group(1)) + 1
            else:
                w = re.match(r"v_\w+ v\[?(\d+)(?::(\d+))?\]?", nxt)
                if w:
                    lo = int(w.group(1))
                    hi = int(w.group(2) or lo)
                    if not (hi < a or lo > b):
                        hazard += 1
                        break
                states += 1
            j += 1
    valu = sum(1 for l in ins if l.startswith("v_") and not l.startswith("v_mfma"))
    scr_loop = 0
    if mf:
        scr_loop = sum(1 for l in ins[longest[0]:longest[-1]] if l.startswith("scratch_"))
    return dict(kernel=name, spilled_vgprs=spills, scratch_reloads_between_output_stores=between, vmcnt0_inside_plane_loop=drains,
                mfma=len(mf), stores=len(stores), store_data_overwritten_next_slot=hazard, scratch_in_plane_loop=scr_loop, valu=valu, v_mov=sum(1 for l in ins if l.startswith("v_mov")), v_pk=sum(1 for l in ins if l.startswith("v_pk_")),
                ds=sum(1 for l in ins if l.startswith("ds_")))


def file_flags(src):
    """the per-file flags of the shipped build (csrc/Makefile: FLAGS_<stem> = ...)"""
    m = re.search(r"^FLAGS_%s\s*=\s*(.*)$" % re.escape(src[:-4]), open(os.path.join(CS, "Makefile")).read(), re.M)
    return (m.group(1).split() if m else []) + os.environ.get("BX_LINT_FLAGS", "").split()      # BX_LINT_FLAGS: variant builds


def main():
    bad = 0
    for src in ("k_wino43.hip", "k_wino43v.hip", "k_wino43m.hip"):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + file_flags(src) + ["-o", out, os.path.join(CS, src)], check=True, stderr=subprocess.DEVNULL)
            asm = open(out).read()
        spills = dict(zip(re.findall(r"^\s+\.name:\s+(_Z\w+)", asm, re.M), [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]))
        for name, ins in kernels(asm).items():
            if "wino43" not in name:
                continue
            short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:48]
            r = report(short, ins, spills.get(name, -1))
            print(r)
            if r["store_data_overwritten_next_slot"]:
                bad += 1
            if src == "k_wino43.hip" and "Li64ELi64" in name and (r["scratch_reloads_between_output_stores"] or r["vmcnt0_inside_plane_loop"]):
                bad += 1

    # k_fps.hip: the resolving wave keeps a round's samples in lanes through v_writelane with the lane select in m0, written in inline asm.
    # m0 is a reserved register: the compiler does not track the clobber, so nothing else in the kernels may use it -- every m0 in the
    # file's assembly must be one of the asm block's own instructions, and the s_mov that loads it must sit right in front of its users.
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + file_flags("k_fps.hip") + ["-o", out, os.path.join(CS, "k_fps.hip")], check=True, stderr=subprocess.DEVNULL)
        lines = [l.split(";")[0].strip() for l in open(out).read().splitlines()]
    lines = [l for l in lines if l and not l.startswith((".", "//"))]
    m0 = [(i, l) for i, l in enumerate(lines) if re.search(r"\bm0\b", l)]
    foreign = [l for _, l in m0 if not re.match(r"(s_mov_b32 m0, s\d+|v_writelane_b32 v\d+, s\d+, m0)$", l)]
    loads = [i for i, l in m0 if l.startswith("s_mov_b32 m0")]
    detached = [i for i in loads if not (lines[i + 1] == "s_nop 0" and all(lines[i + 2 + k].startswith("v_writelane_b32") and lines[i + 2 + k].endswith(", m0") for k in range(5)))]
    print("k_fps_m0 uses=%d loads=%d foreign=%d detached=%d" % (len(m0), len(loads), len(foreign), len(detached)))
    if foreign or detached or len(m0) != 6 * len(loads) or not loads:
        bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
