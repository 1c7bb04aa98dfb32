"""Minimal stand-in for the reference Logger's hot-path surface (logger.py:26-119): ``log``,
``scalar``, ``eval_scalar``, ``image_color``, ``image_grey``, ``pointcloud``, ``model``.
TensorBoard / cv2 / open3d writers are out of scope; what matters for the hot path is that
``scalar`` does NOT synchronise: device tensors are kept as they are and only converted when read
(the reference's ten .item() calls per iteration, logger.py:105-108, would stall every step)."""
import os

import numpy as np
import torch


class Logger(object):
    def __init__(self, root_dir=None, cfg=None, verbose=True):
        self.root_dir = root_dir
        self.cfg = cfg
        self.verbose = verbose
        self._scalars = {}
        self.eval_scalars = {}
        self.images = {}
        if root_dir is not None:
            os.makedirs(os.path.join(root_dir, 'model'), exist_ok=True)

    def log(self, string):
        if self.verbose:
            print(string, flush=True)
        if self.root_dir is not None:
            with open(os.path.join(self.root_dir, 'events.log'), 'a') as f:
                f.write(string + '\n')

    def scalar(self, name, value, step):
        self._scalars.setdefault(name, []).append(value.detach() if torch.is_tensor(value) else value)

    @property
    def scalars(self):
        """name -> list of Python floats (this is where the deferred device->host copies happen)."""
        out = {}
        for k, vals in self._scalars.items():
            where = [i for i, v in enumerate(vals) if torch.is_tensor(v)]
            if where:       # one device->host copy for the tensors logged since the last read; floats stay as they are
                got = torch.stack([vals[i].to(torch.float32).reshape(()) for i in where]).cpu().tolist()
                vals = list(vals)
                for i, g in zip(where, got):
                    vals[i] = g
                self._scalars[k] = vals
            out[k] = [float(v) for v in vals]
        return out

    def eval_scalar(self, name, value, step):
        self.eval_scalars.setdefault(name, []).append(float(value))

    def image_color(self, name, image, step):
        self.images[name] = np.asarray(image)

    def image_grey(self, name, image, step):
        self.images[name] = np.asarray(image)

    def pointcloud(self, pointcloud, step, max_variance=None):
        self.images['pointcloud'] = pointcloud

    def model(self, model, step):
        if self.root_dir is not None:
            torch.save(model.state_dict(), os.path.join(self.root_dir, 'model', '%d.pth' % step))
