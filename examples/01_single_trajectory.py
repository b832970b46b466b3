#!/usr/bin/env python3
"""The reference's own call, unchanged, on the GPU.

    from optimizations import *                      (scripts/drones_pols_generator.py:16)
    pols_coeffs, pc_pols = calculate_trajectory4D(traj_points)   (:58)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drone_path_planning_python_amd.optimizations as optimizations  # noqa: E402

sys.modules["optimizations"] = optimizations          # the alias INTEGRATION.md section 1 describes
from optimizations import Point_time, Waypoint, calculate_trajectory4D  # noqa: E402

traj_points = [
    Point_time(Waypoint(0.0, 0.0, 0.0, 0.0), t=0),
    Point_time(Waypoint(2.0, 2.2, 0.3, 0.0), t=1),
    Point_time(Waypoint(4.0, 8.0, 0.8, 0.0), t=3),
    Point_time(Waypoint(1.0, 2.0, 0.4, 0.5), t=4),
]
pols_coeffs, pc_pols = calculate_trajectory4D(traj_points)
print("segments:", len(pols_coeffs[0]), "durations:", pc_pols[0].time_durations)
print("x(t) piece 1 coefficients (ascending powers):", pols_coeffs[0][1].p.ravel())
for t in (0.0, 0.5, 1.0, 2.0, 3.5, 4.0):
    print(f"t = {t:3.1f}  x = {float(pc_pols[0].eval(t)[0]):8.4f}  y = {float(pc_pols[1].eval(t)[0]):8.4f}")
