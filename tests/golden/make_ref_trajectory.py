#!/usr/bin/env python3
"""Golden vectors from the REFERENCE itself - the one slice of it that runs in the build container.

`safemotions/utils/trajectory_manager.py` imports without pybullet / klimits (everything else of the package needs one of them,
SURVEY.md section 8c).  This script imports that module from /root/reference (read-only, no bytecode written), drives
`TrajectoryManager.reset()` / `is_trajectory_finished(index)` (tm.py:77-108,187-192) and `clip_index` with an obstacle wrapper
argument that only hands back a start state, and writes the results to tests/golden/ref_trajectory.json.  It pins the episode-length
rule of row A16 (`trajectory_length = round(T / dt) + 1`, done when `episode_length >= trajectory_length - 1`) and nothing more:
parity of the distance / safe-range arithmetic stays unpinned.

    python -B tests/golden/make_ref_trajectory.py        # only where /root/reference exists; the fixture is committed
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REFERENCE = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_trajectory.json")


class _StartStateOnly:
    """the constructor argument `obstacle_wrapper`: TrajectoryManager.reset() asks it for the start state (tm.py:194-195)"""

    def get_starting_point_joint_pos_vel_acc(self):
        return [0.0] * 7, None, None


def main():
    sys.path.insert(0, REFERENCE)
    import safemotions.utils.trajectory_manager as tm
    cases = []
    for dt in (0.1, 0.05):
        for duration in (2.0, 3.0, 8.0, 0.25, 1.04):
            m = tm.TrajectoryManager(trajectory_time_step=dt, trajectory_duration=duration, obstacle_wrapper=_StartStateOnly())
            m.reset()
            n = int(m.trajectory_length)
            finished = [bool(m.is_trajectory_finished(i)[0]) for i in range(n + 2)]
            reasons = sorted({m.is_trajectory_finished(i)[1] for i in range(n + 2)} - {None})
            cases.append(dict(trajectory_time_step=dt, trajectory_duration=duration, trajectory_length=n, finished=finished,
                              reasons=reasons, first_finished_index=finished.index(True)))
    clip = [[i, n, int(tm.clip_index(i, n))] for n in (1, 3, 21) for i in (-25, -4, -3, -1, 0, 1, 2, 3, 20, 21, 40)]
    json.dump(dict(source="safemotions/utils/trajectory_manager.py (reference, imported as is)",
                   termination_trajectory_length=int(tm.TERMINATION_TRAJECTORY_LENGTH), cases=cases, clip_index=clip),
              open(OUT, "w"), indent=1)
    print("wrote", OUT, "-", len(cases), "cases")


if __name__ == "__main__":
    main()
