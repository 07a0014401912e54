"""Telling the native modules that an optimizer stepped.

``HiFiGANGenerator`` / ``GBlockGenerator`` / the discriminators hand their parameters to libhificar when the tensors' version counters say they
changed.  torch's FUSED optimizers (``torch.optim.Adam(..., fused=True)``) update in place WITHOUT bumping ``Parameter._version`` (checked on
torch 2.10), so a module watching versions alone would run its next forward on the weights of the step before.  ``watch(module)`` — called by the
modules' constructors — registers ONE process-wide optimizer post-step hook (``torch.optim.optimizer.register_optimizer_step_post_hook``): after
every ``optimizer.step()``, every live watched module that owns one of that optimizer's parameters gets ``invalidate_parameters()``.  Nothing for
the user to wire (round 3 relied on hooks the Trainer registered; any other training loop silently used stale weights).
"""

import weakref

_watched = weakref.WeakSet()
_handle = None


def _module_parameters(module):
    lst = getattr(module, "_plist", None)
    return lst() if callable(lst) else list(module.parameters())


def _on_step(optimizer, args, kwargs):
    if not _watched:
        return
    ids = {id(p) for group in optimizer.param_groups for p in group["params"]}
    for m in list(_watched):
        if any(id(p) in ids for p in _module_parameters(m)):
            m.invalidate_parameters()


def watch(module):
    """After every ``optimizer.step()`` of an optimizer holding one of ``module``'s parameters: ``module.invalidate_parameters()``."""
    global _handle
    _watched.add(module)
    if _handle is None:
        from torch.optim import optimizer as _opt

        _handle = _opt.register_optimizer_step_post_hook(_on_step)
