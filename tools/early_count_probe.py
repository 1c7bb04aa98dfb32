#!/usr/bin/env python3
"""Run-to-run spread of a 3-step binned training run against the early / late count pass (tests/test_gpu_table_update.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_table_update as T


class MP(object):
    def setattr(self, obj, name, val):
        setattr(obj, name, val)

    def setenv(self, k, v):
        os.environ[k] = v


def metrics(a, b):
    (p1, m1, v1), (p0, m0, v0) = a, b
    return [round(float((m1 - m0).norm() / m0.norm()), 6), round(float((v1 - v0).norm() / v0.norm()), 6),
            float(((m1 - m0).abs() > 1e-4 * float(m0.abs().max())).float().mean()),
            float(((p1 - p0).abs() > 1e-5).float().mean())]


runs = {(e, i): T._early_run(e, MP(), False) for e in (True, False) for i in range(3)}
for i in range(3):
    print('early vs late', i, metrics(runs[(True, i)], runs[(False, i)]))
print('late vs late  ', metrics(runs[(False, 0)], runs[(False, 1)]), metrics(runs[(False, 1)], runs[(False, 2)]))
print('early vs early', metrics(runs[(True, 0)], runs[(True, 1)]), metrics(runs[(True, 1)], runs[(True, 2)]))
