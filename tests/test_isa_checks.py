"""CPU: the disassembly checks of tools/isa_hazard_check.py hold for the library as built (they also
run inside __graft_entry__.build()), and the checker itself notices a violation when the required
distance is raised by one (a self-test: the carry groups sit exactly two wait states apart)."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOOL = os.path.join(ROOT, "tools", "isa_hazard_check.py")
OBJS = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "fp_inst_*.o")))


@pytest.mark.skipif(not OBJS, reason="objects are built by __graft_entry__.build()")
def test_no_sgpr_hazard_and_no_accvgpr_moves_in_the_mfma_stream():
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 SGPR hazards, 0 accvgpr moves" in r.stdout


@pytest.mark.skipif(not OBJS, reason="objects are built by __graft_entry__.build()")
def test_checker_detects_a_shorter_distance():
    one = [o for o in OBJS if o.endswith("fp_inst_8.o")] or OBJS[:1]
    r = subprocess.run([sys.executable, TOOL] + one, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ISA_CHECK_WAIT_STATES="3"))
    assert r.returncode == 1 and "HAZARD" in r.stdout
