"""Properties of the BUILT device code that the source can only ask for (read out of the library's gfx950
code object, no GPU needed).

* The covariance pass hands a tile's row of partial sums to the merging workgroup through a ticket counter
  (csrc/swimmer_kernels.hip, moments_tile).  The row is written with agent-scope write-through stores and the
  ticket must not become visible before those stores have completed: every wave waits `s_waitcnt vmcnt(0)`
  before the barrier behind which thread 0 takes the ticket.  A workgroup-scope release fence does not emit
  that wait on gfx950 (round 3 relied on it), so the test looks for the instruction itself: between the last
  global store ahead of the ticket's `global_atomic_add` and the atomic there must be a `vmcnt(0)` wait.
* The product contains no matrix instruction (north_star: "no MFMA -- this is a bandwidth / latency problem").
"""
import os
import re
import shutil

import pytest

from test_loop_placement import LIB, LLVM, _disassemble

needs_tools = pytest.mark.skipif(not (os.path.exists(LIB) and shutil.which(f"{LLVM}/llvm-objdump")),
                                 reason="needs the built library and the ROCm llvm tools")

# every kernel that contains moments_tile: the standalone pass and the rollout kernels it rides along in
TICKET_KERNELS = [
    "traj_moments_kernelILi6E", "traj_moments_kernelILi8ELi128E", "traj_moments_kernelILi8ELi256E",
    "traj_moments_kernelILi10E", "traj_moments_kernelILi12E", "traj_moments_kernelILi14ELi256E",
    "traj_moments_kernelILi16E", "traj_moments_kernelILi18E",
    "rollout_oct3_kernelILb1ELb1ELb1E", "rollout_quad3_kernelILb1ELb1ELb1E",
    "rollout_row_kernelILi4ELb1ELb1ELb1E", "rollout_row_kernelILi5ELb1ELb1ELb1E",
    "rollout_row_kernelILi6ELb1ELb1ELb1E", "rollout_row_kernelILi7ELb1ELb1ELb1E",
    "rollout_row_kernelILi8ELb1ELb1ELb1E",
]


def _kernels(lines, fragment):
    """[(symbol, [instruction text])] of every kernel whose mangled name contains `fragment`."""
    out = []
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.endswith(">:") and fragment in l and l.startswith("0000"):
            body = []
            i += 1
            while i < len(lines) and not lines[i].startswith("0000"):
                m = re.match(r"\s+(\S.*?)\s*//\s*[0-9A-Fa-f]+:", lines[i])
                if m:
                    body.append(m.group(1))
                i += 1
            out.append((l, body))
        else:
            i += 1
    return out


@needs_tools
def test_ticket_is_taken_after_the_tile_rows_stores_have_completed(tmp_path):
    lines = _disassemble(tmp_path)
    seen = 0
    for fragment in TICKET_KERNELS:
        ks = _kernels(lines, fragment)
        assert ks, (fragment, "not in the library")
        for sym, body in ks:
            tickets = [i for i, x in enumerate(body) if re.match(r"global_atomic_add(_u32)?\s", x)]
            assert len(tickets) == 1, (sym, "expected exactly one ticket atomic", tickets)
            i = tickets[0]
            stores = [j for j in range(i) if re.match(r"(global|buffer|flat)_store", body[j])]
            assert stores, (sym, "no store ahead of the ticket?")
            between = body[stores[-1] + 1:i]
            assert any(x.startswith("s_waitcnt") and "vmcnt(0)" in x for x in between), \
                (sym, "no s_waitcnt vmcnt(0) between the tile row's last store and the ticket", between[-12:])
            if any(x.startswith("s_barrier") for x in body):   # one-wave workgroups have none (the compiler drops it)
                assert any(x.startswith("s_barrier") for x in between), (sym, "the waves must meet before the ticket")
            seen += 1
    assert seen >= len(TICKET_KERNELS)


@needs_tools
def test_no_matrix_instructions_in_the_product(tmp_path):
    lines = _disassemble(tmp_path)
    assert not [l for l in lines if re.match(r"\s+v_(s?mfma|wmma)", l)]
