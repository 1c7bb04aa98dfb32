"""PSNR delta between the HIP path and the CPU oracle after an identical training budget ("matching PSNR on
the synthetic scene"): same initial weights, same ray batches, same stratification noise, same Adam groups, 60
iterations of 1024 rays on 20 views of the analytic scene; then both render one training view.  Bar: 0.05 dB."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_psnr_delta_after_identical_training():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import psnr_run
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r = psnr_run.delta_run(res=64, n_img=20, L=8, T=15, iters=60, n_rays=1024)
    assert r['psnr_hip'] > 15.0, r                      # the budget is enough to learn something
    assert abs(r['psnr_delta']) <= 0.05, r
    assert r['max_abs_loss_diff'] <= 5e-4, r


def test_psnr_delta_on_the_binned_table_update():
    """The same comparison on a table whose hashed levels take the DEFAULT update of the metric configuration
    (nerf_table_bin_* + nerf_table_update_binned, csrc/table_update.hip: 47 % of the metric step) and through the route
    Trainer.train_step takes (render_loss: compositing + losses fused): config_base's 20 levels with T = 2^19, 40 iterations
    of 1024 rays.  Quality check of the reference: trainer.py:95-101, helpers.py:11-13.  Bar: 0.05 dB."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import psnr_run
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r = psnr_run.delta_run(res=64, n_img=20, L=20, T=19, iters=40, n_rays=1024)
    assert len(r['binned_levels']) >= 8 and r['fused_tail'], r      # the hashed levels 9..19 and the big dense ones
    assert r['skipped_steps'] == 0, r
    assert r['psnr_hip'] > 13.0, r
    assert abs(r['psnr_delta']) <= 0.05, r
    assert r['max_abs_loss_diff'] <= 5e-4, r


def test_psnr_delta_with_the_half_precision_table():
    """`precision: float16` (tests/test_gpu_half_table.py): the same comparison with both sides gathering from the fp16 image
    of their fp32 master table, the HIP side's image written by the binned update pass itself.  Bar: 0.05 dB."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import psnr_run
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r = psnr_run.delta_run(res=64, n_img=20, L=20, T=19, iters=30, n_rays=1024, precision='float16')
    assert len(r['binned_levels']) >= 8 and 'float16' in r['config'], r
    assert r['skipped_steps'] == 0, r
    assert r['psnr_hip'] > 12.0, r
    assert abs(r['psnr_delta']) <= 0.05, r
    assert r['max_abs_loss_diff'] <= 5e-4, r
