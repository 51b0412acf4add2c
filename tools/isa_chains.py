"""python tools/isa_chains.py <kernel name substring> file.s [...]: innermost loops of a kernel's gfx950 ISA that hold a vector-memory
load AND wait for every outstanding one (s_waitcnt vmcnt(0)) -- i.e. one memory round trip per iteration, the pattern behind round 5's
finds (the Adam replay's per-step constants, row_bwd's slice sums, the dx reduction's mask re-read).  Prints kernel, loop label,
lines, loads, waits.  (hipcc ... -save-temps=obj leaves the .s files next to the object.)"""
import re
import sys

pat = sys.argv[1]
for path in sys.argv[2:]:
    name = None
    lines = open(path).read().split("\n")
    i = 0
    kern = {}
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            name = m.group(1)
            kern[name] = []
        elif name is not None:
            kern[name].append(ln)
            if "s_endpgm" in ln:
                name = None
    for k, body in kern.items():
        if pat not in k:
            continue
        # blocks by label; loop membership from the compiler's comments ("in Loop: Header=BBx_y Depth=n" / "Loop Header")
        label, blocks = "entry", {}
        header_of = {}
        for ln in body:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                label = m.group(1)
            blocks.setdefault(label, []).append(ln)
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", ln)
            if m:
                header_of[label] = "." + "L" + m.group(1)
            if "Inner Loop Header" in ln or "This Loop Header" in ln:
                header_of.setdefault(label, label)
        loops = {}
        for lab, hdr in header_of.items():
            loops.setdefault(hdr, []).extend(blocks.get(lab, []))
        for hdr, ls in loops.items():
            loads = [l.strip() for l in ls if re.search(r"\b(global_load|flat_load|buffer_load)", l)]
            waits = [l for l in ls if "vmcnt(0)" in l]
            if loads and waits:
                n = len([l for l in ls if l.strip() and not l.strip().startswith(";")])
                print(f"{k[:90]}  loop {hdr}: {n} lines, {len(loads)} loads, {len(waits)} full waits   e.g. {loads[0][:60]}")
