"""ISA guards: kernels whose correctness rests on what the compiler put between two instructions it cannot relate -- inline-asm loads
awaited by hand-counted `s_waitcnt vmcnt(n)` (csrc/attention.hip: mha_bwd1s_kernel; csrc/gemm_nt.hip: the ticket walk's mailbox word).
The checks read the device assembly `-save-temps` leaves beside the object of the SAME compile that produced the shipped code
(vipant_amd/lib/obj/<name>-hip-amdgcn-amd-amdhsa-gfx950.s) and raise AssertionError.  `vipant_amd.build` runs them after compiling and
fails the build if one does not hold (a toolchain that moves or spills one of these registers must not produce a library);
tests/test_abi_cpu.py runs the same functions.  VIPANT_GEMM_VARIANT bit 22 (4194304) selects the static tile walk at run time -- the
fallback should a future hipcc break the ticket walk's guard."""
from __future__ import annotations

import os
import re


def check_attention(path: str) -> None:
    """mha_bwd1s_kernel requests the NEXT problem's V fragments with inline-asm `global_load_dwordx4` into AGPRs (`"=&a"`) and awaits
    them a stage later with a hand-counted `s_waitcnt vmcnt(22)` (62 in the kernel that also emits e4m3 dK / dV).  Between the first
    of the ten loads and the counted wait no instruction may name any of the loaded AGPRs (nor the AGPR of the ticket draw)."""
    assert os.path.exists(path), path
    text = open(path).read()
    found = re.findall(r"^(_ZN\S*mha_bwd1s_kernelILi20ELb([01])E\S*):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M)
    assert sorted(f[1] for f in found) == ["0", "1"], [f[0] for f in found]       # the plain kernel and the one that also emits e4m3 dK / dV
    for name, q8, code in found:
        _check_bwd1s_isa(code.split("\n"), 62 if q8 == "1" else 22)


def _check_bwd1s_isa(lines, nwait):
    waits = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt\s+vmcnt\(%d\)" % nwait, l)]
    assert len(waits) == 1, (nwait, waits)

    def agprs(line):
        regs = set()
        for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", line):
            regs.update(range(int(lo), int(hi) + 1))
        regs.update(int(r) for r in re.findall(r"\ba(\d+)\b", line))
        return regs

    loads = [i for i in range(waits[0]) if re.search(r"global_load_dwordx4\s+a\[", lines[i])]
    group = loads[-10:]
    assert len(group) == 10 and group[-1] - group[0] < 200, (len(loads), group)      # the in-loop v_load(pn): ten loads back to back
    loaded = set().union(*(agprs(lines[i].split(",")[0]) for i in group))
    assert len(loaded) == 40, sorted(loaded)
    # the ticket of the problem after the next (round 5): one asm atomic just in front of those loads, result in an AGPR, same wait
    draws = [i for i in range(group[0]) if re.search(r"global_atomic_add\s+a\d+,", lines[i])]
    assert len(draws) == 1 and group[0] - draws[0] < 60, (draws, group[0])
    loaded |= agprs(lines[draws[0]].split(",")[0])
    assert len(loaded) == 41, sorted(loaded)
    # between the loads and the counted wait: exactly the stores the count assumes (2 of dQ + 20 of dK / dV [+ 40 e4m3 / scale stores])
    stores = [i for i in range(group[-1], waits[0]) if re.search(r"^\s*(buffer|global)_store", lines[i])]
    assert len(stores) == nwait, (nwait, len(stores))
    for i in range(draws[0] + 1, waits[0]):
        if i in group or lines[i].lstrip().startswith(";"):
            continue
        hit = agprs(lines[i]) & loaded
        assert not hit, (i, lines[i].strip(), sorted(hit))


def check_gemm_nt(path: str) -> None:
    """Every wave of a ticket-walk NT kernel requests its mailbox word with an inline-asm `global_load_dword` BEHIND a tile's epilogue
    stores and reads the register in the next tile's bias round trip, after at least three K-tiles of hand-counted vmcnt waits.  For
    every ticket instantiation: one asm load; in program text the register's last mention is that load (the loop's back edge follows);
    it is read by a `v_readfirstlane_b32`; and no instruction anywhere COPIES it (the allocator may reuse the register as an
    arithmetic temporary between the read and the next load)."""
    assert os.path.exists(path), path
    text = open(path).read()
    kernels = re.findall(r"^(_ZN\S*gemm_nt_pp_kernelILi\d+ELi\d+ELi2ELi0ELb1E\S*):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) >= 8, len(kernels)
    for name, code in kernels:
        lines = [l.split(";")[0] for l in code.split("\n")]
        raw = code.split("\n")
        loads = [i for i, l in enumerate(lines) if re.match(r"\s*global_load_dword\s+v\d+,\s*v\[\d+:\d+\],\s*off\s*$", l)
                 and "ASMSTART" in raw[i - 1]]
        assert len(loads) == 1, (name, loads)
        reg = re.match(r"\s*global_load_dword\s+(v\d+),", lines[loads[0]]).group(1)
        uses = [i for i, l in enumerate(lines) if re.search(r"\b%s\b" % reg, l)]
        assert uses[-1] == loads[0], (name, reg, [lines[i].strip() for i in uses[-3:]])
        assert any(re.match(r"\s*v_readfirstlane_b32\s+s\d+,\s*%s\b" % reg, lines[i]) for i in uses), (name, reg)
        # a copy / spill of the register is a violation unless the allocator has given the register a NEW value first (a write to it,
        # other than the pre-loop zero, earlier in program text: the loop body runs from the bias round trip's read to the asm load)
        init = [i for i in uses if re.match(r"\s*v_mov_b32_e32\s+%s,\s*0\s*$" % reg, lines[i])]
        writes = [i for i in uses if i not in init and i != loads[0] and re.match(r"\s*v_\w+\s+%s\b" % reg, lines[i])
                  and not re.match(r"\s*v_(cmp|readfirstlane|readlane)", lines[i])]
        copies = [i for i in uses
                  if re.match(r"\s*(v_mov_b32_e32\s+v\d+|v_accvgpr_write_b32\s+a\d+|v_writelane_b32\s+v\d+),\s*%s\b" % reg, lines[i])
                  or re.match(r"\s*(scratch_store|global_store|buffer_store|ds_write)\S*\s.*\b%s\b" % reg, lines[i])]
        bad = [lines[i].strip() for i in copies if not any(w < i for w in writes)]
        assert not bad, (name, reg, bad)


CHECKS = {"attention.hip": check_attention, "gemm_nt.hip": check_gemm_nt}
