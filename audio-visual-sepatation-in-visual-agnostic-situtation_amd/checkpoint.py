"""Checkpoint files with the reference's names and keys (reference: main.py:506-533, 606-621,
780-797).

``checkpoint(nets, history, itera, args)`` writes ``{ckpt}/sound_latest.pth``, ``frame_latest.pth`` and
``history_latest.pth`` (plain ``state_dict()``s / the history dict, loadable by the reference) and the ``*_best.pth``
pair when ``-history['val_ao']['si_sdr'][-1]`` improves on ``args.best_err``.  The reference forgets the optimizer
and the iteration counter, so a resumed run restarts the momentum; ``optimizer`` (a FlatSGD) adds
``optim_latest.pth`` for that — an extra file the reference ignores.

The SoP++ three-stage form (``nets`` = (sound, frame, synthesizer, attention module), SoP++/main.py:599-631) also writes
``synthesizer_{latest,best}.pth`` and ``net_pit_latest.pth`` like the reference, plus ``net_pit_best.pth`` (the reference
writes ``net_pit_latest.pth`` twice at :630-631 and never reloads the module, :913-921; here ``--load_ckpt`` and eval
mode restore it, so that a resumed or evaluated run has the attention module it was saved with).
"""
import os

import torch


def _save(obj, path):
    """torch.save to a temporary file + os.replace: a kill mid-save never corrupts the only resume point."""
    tmp = path + ".tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def _cpu_state(net):
    return {k: v.detach().cpu() for k, v in net.state_dict().items()}


def checkpoint(nets, history, itera, args, optimizer=None):
    print("Saving checkpoints at {} iterations.".format(itera))
    if len(nets) not in (2, 4):
        raise ValueError("checkpoint() takes (sound, frame) or (sound, frame, synthesizer, attention module)")
    net_sound, net_frame = nets[:2]
    extra = list(zip(("synthesizer", "net_pit"), nets[2:]))
    os.makedirs(args.ckpt, exist_ok=True)
    path = lambda what, suffix: os.path.join(args.ckpt, "{}_{}".format(what, suffix))   # noqa: E731
    _save(history, path("history", "latest.pth"))
    _save(_cpu_state(net_sound), path("sound", "latest.pth"))
    _save(_cpu_state(net_frame), path("frame", "latest.pth"))
    for what, net in extra:
        _save(_cpu_state(net), path(what, "latest.pth"))
    if optimizer is not None:
        _save({"itera": itera, "state": optimizer.state_dict()}, path("optim", "latest.pth"))
    cur_err = -history["val_ao"]["si_sdr"][-1]
    if cur_err < args.best_err:
        args.best_err = cur_err
        _save(_cpu_state(net_sound), path("sound", "best.pth"))
        _save(_cpu_state(net_frame), path("frame", "best.pth"))
        for what, net in extra:
            _save(_cpu_state(net), path(what, "best.pth"))


def resume_paths(args, best=False, three_stage=False):
    """The weight paths the reference derives for --resume ('latest') and for eval mode ('best'), main.py:780-791;
    `three_stage` adds the synthesizer's (SoP++/main.py:913-921) and the attention module's."""
    suffix = "best.pth" if best else "latest.pth"
    names = ("sound_", "frame_") + (("synthesizer_", "net_pit_") if three_stage else ())
    return tuple(os.path.join(args.ckpt, n + suffix) for n in names)


def load_optimizer(optimizer, args):
    p = os.path.join(args.ckpt, "optim_latest.pth")
    if not os.path.exists(p):
        return 0
    blob = torch.load(p, map_location="cpu")
    optimizer.load_state_dict(blob["state"])
    return int(blob["itera"])
