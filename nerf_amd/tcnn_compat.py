"""Conversion between this package's flat MLP parameters and tiny-cuda-nn's FullyFusedMLP layout, for state_dicts saved by
the reference (logger.py:101-103 dumps ``model.state_dict()``: keys ``encoder.params``, ``sigma_net.params``,
``color_net.params``, ``latent_emb``).

PARITY UNPINNED.  tinycudann is not importable here and the reference ships no checkpoint, so nothing below could be checked
against a file written by the reference; it follows tiny-cuda-nn's published layout: every layer is a
row-major [out, in] matrix, layers concatenated in order, the FIRST layer's input width and the LAST layer's output width
padded with zeros to a multiple of 16, parameters possibly stored in half precision.  The hash-grid table
(``encoder.params``: levels one after the other, F floats per entry) and ``latent_emb`` need no conversion beyond the dtype.
What IS tested: strip(pad(x)) == x, pad(strip(y)) == y on the padded positions being zero, and a padded state_dict loading
into a NeRFNetwork that then evaluates exactly like the one it was exported from (tests/test_tcnn_compat.py).
"""
import torch


def _pad16(n):
    return (n + 15) // 16 * 16


def mlp_widths(n_in, n_out, n_neurons, n_hidden_layers):
    return [n_in] + [n_neurons] * n_hidden_layers + [n_out]


def padded_numel(widths):
    w = list(widths)
    w[0], w[-1] = _pad16(w[0]), _pad16(w[-1])
    return sum(w[i + 1] * w[i] for i in range(len(w) - 1))


def strip_padding(flat, widths):
    """tiny-cuda-nn FullyFusedMLP parameters (padded, any float dtype) -> this package's flat fp32 layout."""
    flat = torch.as_tensor(flat).detach().reshape(-1).to(torch.float32)
    wp = list(widths)
    wp[0], wp[-1] = _pad16(wp[0]), _pad16(wp[-1])
    if flat.numel() != padded_numel(widths):
        raise ValueError('expected %d padded parameters for layer widths %s, got %d'
                         % (padded_numel(widths), list(widths), flat.numel()))
    out, o = [], 0
    for i in range(len(widths) - 1):
        rows_p, cols_p = wp[i + 1], wp[i]
        layer = flat[o:o + rows_p * cols_p].view(rows_p, cols_p)
        o += rows_p * cols_p
        out.append(layer[:widths[i + 1], :widths[i]].reshape(-1))
    return torch.cat(out)


def add_padding(flat, widths, dtype=torch.float32):
    """This package's flat layout -> tiny-cuda-nn's padded one (zeros in the padding)."""
    flat = torch.as_tensor(flat).detach().reshape(-1).to(torch.float32)
    wp = list(widths)
    wp[0], wp[-1] = _pad16(wp[0]), _pad16(wp[-1])
    out, o = [], 0
    for i in range(len(widths) - 1):
        rows, cols = widths[i + 1], widths[i]
        layer = torch.zeros(wp[i + 1], wp[i], dtype=torch.float32, device=flat.device)
        layer[:rows, :cols] = flat[o:o + rows * cols].view(rows, cols)
        o += rows * cols
        out.append(layer.reshape(-1))
    if o != flat.numel():
        raise ValueError('parameter count does not match layer widths %s' % list(widths))
    return torch.cat(out).to(dtype)


def from_reference_state_dict(state_dict, model):
    """A state_dict saved by the reference (tinycudann modules) -> one that ``model.load_state_dict`` accepts.
    Tensors that already have this package's sizes pass through unchanged."""
    out = {}
    for key, val in state_dict.items():
        val = torch.as_tensor(val)
        mod = {'sigma_net.params': getattr(model, 'sigma_net', None), 'color_net.params': getattr(model, 'color_net', None)}.get(key)
        if mod is not None and val.numel() != mod.params.numel():
            val = strip_padding(val, mod.widths)
        out[key] = val.to(torch.float32).reshape(model.state_dict()[key].shape) if key in model.state_dict() else val
    return out


def to_reference_state_dict(model, dtype=torch.float32):
    """The reverse: this model's parameters in tiny-cuda-nn's padded layout."""
    out = {}
    for key, val in model.state_dict().items():
        mod = {'sigma_net.params': getattr(model, 'sigma_net', None), 'color_net.params': getattr(model, 'color_net', None)}.get(key)
        out[key] = add_padding(val, mod.widths, dtype) if mod is not None else val.detach().clone()
    return out
