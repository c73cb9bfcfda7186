"""Pipeline stages as PROCESSES (one stage each, both on the one GPU): the load-time hand-off of the meta rows inside
``load_weights`` and a step's hand-off run over a torch.distributed group (gloo, host copies), as they would between the
ranks of a pipeline group under an engine."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_pipeline_stage_processes_equal_the_whole_model(tmp_path):
    port = str(29600 + os.getpid() % 300)
    procs, results = [], []
    for rank in range(2):
        results.append(str(tmp_path / f"rank{rank}.txt"))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "pp_two_process_worker.py"), str(rank), "2", port,
                                       results[-1]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert [p.returncode for p in procs] == [0, 0], "\n".join(outs)
    assert open(results[0]).read() == "sent" and open(results[1]).read() == "equal"
