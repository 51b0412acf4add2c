"""python tools/isa_chains.py <kernel name substring> file.s [...]: loops of a kernel's gfx950 ISA that hold a vector-memory
load AND wait for every outstanding one (s_waitcnt vmcnt(0)) -- i.e. one memory round trip per iteration, the pattern behind round 5's
finds (the Adam replay's per-step constants, row_bwd's slice sums, the dx reduction's mask re-read, the loss rows' multiplicities).
Prints kernel, loop label, lines, loads, waits.  (hipcc ... -save-temps=obj, or --cuda-device-only -S, writes the .s files.)
tests/test_host_logic.py uses kernels() / loops() to keep the replay loop of the row-lazy Adam free of vector loads."""
import re
import sys

VMEM_LOAD = re.compile(r"\b(global_load|flat_load|buffer_load|scratch_load)")


def kernels(path):
    """{mangled kernel name: [ISA lines]} of one assembly file."""
    name, out = None, {}
    for ln in open(path).read().split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None:
            out[name].append(ln)
            if "s_endpgm" in ln:
                name = None
    return out


def loops(body):
    """{loop header label: [ISA lines of the blocks the compiler's comments place in that loop]} (innermost membership)."""
    label, blocks, header_of = "entry", {}, {}
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label = m.group(1)
        blocks.setdefault(label, []).append(ln)
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", ln)
        if m:
            header_of[label] = ".L" + m.group(1)
        if "Inner Loop Header" in ln or "This Loop Header" in ln:
            header_of.setdefault(label, label)
    out = {}
    for lab, hdr in header_of.items():
        out.setdefault(hdr, []).extend(blocks.get(lab, []))
    return out


def main(pat, paths):
    for path in paths:
        for k, body in kernels(path).items():
            if pat not in k:
                continue
            for hdr, ls in loops(body).items():
                loads = [l.strip() for l in ls if VMEM_LOAD.search(l)]
                waits = [l for l in ls if "vmcnt(0)" in l]
                if loads and waits:
                    n = len([l for l in ls if l.strip() and not l.strip().startswith(";")])
                    print(f"{k[:90]}  loop {hdr}: {n} lines, {len(loads)} loads, {len(waits)} full waits   e.g. {loads[0][:60]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
