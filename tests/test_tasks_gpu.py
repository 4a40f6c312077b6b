"""Batched task generation on the device (flingbot_amd/tasks.py, SURVEY.md 8f row f4 generator half) against the tasks the
REFERENCE's generate_randomization produced on the oracle (tests/golden/task_golden.npz), bit for bit."""
import pytest

pytestmark = pytest.mark.gpu


def test_device_task_generator_matches_reference_golden(gpu_required):
    from fling_helpers import check_tasks_against_golden
    from flingbot_amd import sim as fsim

    check_tasks_against_golden(lambda n: fsim.FlingSim(n_envs=n, solver=0))
