"""Training / evaluation driver on the MI355X path (the loop of the reference's main.py:572-763, 768-801).

    python -m avsep_amd.train --id run1 --av_list_train data/train_av.csv --ao_list_train data/train_ao.csv \\
        --list_val data/val.csv --arch_sound unet7 --arch_frame resnet18dilated --fusion_type hidsep ... (arguments.py)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m avsep_amd.train ...   # one process per GPU

Same flags, same schedule and same bookkeeping as the reference:

* `av_ao_schedule`: audio-visual step when `i % iter_per_av == 0` (or `i < num_fsteps` with `start_av_first`), else
  audio-only (main.py:578-581); the two loaders are cycled independently;
* `history` with the reference's keys (`train`, `train_ao`, `train_av`, `val_av`, `val_ao`), printed every `disp_iter`;
  the audio-visual running error excludes `match_weight * match_loss` like main.py:718;
* every `eval_iter`: `evaluate()` on the validation list with and without vision, then `checkpoint()` (best model by
  the audio-only SI-SDR); learning rates drop x0.1 at `lr_steps`; `--mode eval` loads `*_best.pth` and only evaluates.

Different on purpose: one process per GPU (torch.distributed + RCCL; each rank draws its own shard of every global
batch) instead of nn.DataParallel; batches carry waveforms and the STFT runs on the GPU; the metrics (SI-SDR and
BSS-eval SDR / SIR / SAR, evaluate.py + bss_eval.py) are computed on the device; no HTML visualisation.
"""
import os
import random
import time

import torch

from . import checkpoint as ckpt
from . import dataset, dp, evaluate as ev, sopp
from .arguments import ArgParser
from .models import ModelBuilder
from .net_wrapper import NetWrapper, adjust_learning_rate, create_optimizer, train_step


def av_ao_schedule(i, args):
    """True -> audio-visual step (main.py:578-581)."""
    if args.start_av_first:
        return i % args.iter_per_av == 0 or i < args.num_fsteps
    return i % args.iter_per_av == 0 and i > args.num_fsteps


class _Cycle:
    """`next(it)`, restarting the loader when it is exhausted (main.py:585-598)."""

    def __init__(self, loader):
        self.loader, self.it = loader, iter(loader)

    def next(self):
        try:
            return next(self.it)
        except StopIteration:
            self.it = iter(self.loader)
            return next(self.it)


def new_history():
    val = lambda: {"iter": [], "err": [], "sdr": [], "sir": [], "sar": [], "si_sdr": []}   # noqa: E731
    return {"train": {"iter": [], "err": []}, "train_ao": {"iter": [], "err": []}, "train_av": {"iter": [], "err": []},
            "val_av": val(), "val_ao": val()}


def evaluate(wrapper, loader, history, itera, args, use_vis, device, world=1):
    """main.py:421-503 without the visualisation: mean loss / match loss / SI-SDR / SDR over the validation list."""
    print("Evaluating at {} iterations...".format(itera))
    wrapper.eval()
    tot = torch.zeros(7, dtype=torch.float64, device=device)      # loss, match, si_sdr, sdr, sir, sar, batches
    with torch.no_grad():
        for host in loader:
            batch = dataset.to_device(host, device)
            err, outputs = wrapper.forward(batch, args, use_vis)
            m = ev.calc_metrics(batch, outputs, args, wrapper.stft_plan)
            match = outputs["match_loss"].mean() if use_vis else err.new_zeros(())
            tot += torch.stack([err.mean().double(), match.double(), m["si_sdr_mean"].double(), m["sdr_mean"].double(),
                                m["sir_mean"].double(), m["sar_mean"].double(), tot.new_ones(())])
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(tot)
    loss, match, si_sdr, sdr, sir, sar = (tot[:6] / tot[6].clamp_min(1)).tolist()
    print("[Eval Summary] iterations: {}, Loss: {:.4f}, Loss_match: {:.4f}, SI-SDR: {:.4f}, SDR: {:.4f}, SIR: {:.4f}, "
          "SAR: {:.4f}".format(itera, loss, match, si_sdr, sdr, sir, sar))
    h = history["val_av" if use_vis else "val_ao"]
    h["iter"].append(itera); h["err"].append(loss); h["sdr"].append(sdr); h["si_sdr"].append(si_sdr)
    h["sir"].append(sir); h["sar"].append(sar)
    torch.set_grad_enabled(True)


def main(args):
    rank, world, device = dp.init_from_env()
    args.device = device
    args.batch_size = args.batch_size_per_gpu                      # per process; the global batch is world x this
    random.seed(args.seed + rank)
    torch.manual_seed(args.seed)                                   # identical replicas
    builder = ModelBuilder()
    net_frame = builder.build_frame(arch=args.arch_frame, fc_dim=args.vis_channels, pool_type=args.img_pool,
                                    weights=args.weights_frame)
    three_stage = getattr(args, "train_steps", None) is not None
    if not three_stage:
        net_sound = builder.build_sound(arch=args.arch_sound, fc_dim=args.num_channels, weights=args.weights_sound,
                                        fusion_type=args.fusion_type, att_type=args.att_type)
    if three_stage:
        # SoP++/main.py:722-747: basis U-Net (extra_size = num_channels), synthesizer, attention module; the AV steps
        # walk through the three stages of SoP++/main.py:670-688 by iteration number
        from .models.attention_net import get_attmodule
        net_sound = builder.build_sound(arch=args.arch_sound, fc_dim=args.num_channels, weights=args.weights_sound,
                                        extra_size=args.num_channels)
        net_syn = builder.build_synthesizer(arch=args.arch_synthesizer, fc_dim=args.num_channels, weights=args.weights_synthesizer)
        net_pit = get_attmodule(args)(att_type=args.att_type)
        pit_path = getattr(args, "weights_net_pit", "")
        if pit_path and not os.path.exists(pit_path) and pit_path.endswith("net_pit_best.pth"):
            # a checkpoint directory written by the reference holds net_pit_latest.pth only (SoP++/main.py:630-631)
            alt = pit_path[:-len("best.pth")] + "latest.pth"
            pit_path = alt if os.path.exists(alt) else ""
            print("net_pit_best.pth is missing: " + ("using net_pit_latest.pth" if pit_path else "attention module keeps its init"))
        if pit_path:
            print("Loading weights for net_pit")
            net_pit.load_state_dict(torch.load(pit_path, map_location="cpu"))
        nets = (net_sound.to(device), net_frame.to(device), net_syn.to(device), net_pit.to(device))
        wrapper = sopp.NetWrapper(nets, builder.build_criterion(arch=args.loss, use_pit=True), builder.build_criterion(arch=args.loss))
        optimizer = sopp.create_optimizer(nets, args, world_size=world)
    else:
        nets = (net_sound.to(device), net_frame.to(device))
        wrapper = NetWrapper(nets, builder.build_criterion(arch=args.loss, use_pit=True), builder.build_criterion(arch=args.loss))
        optimizer = create_optimizer(nets, args, world_size=world)
    torch.manual_seed(args.seed + 1 + rank)                        # rank-local data order and AO swaps

    loader_val = dataset.make_loader(args.list_val, args, "val", args.batch_size, False, workers=min(4, args.workers))
    history = new_history()
    start_i = 0
    if args.load_ckpt:
        print("Recovered from history.")
        history = torch.load(os.path.join(args.ckpt, "history_latest.pth"))
        start_i = ckpt.load_optimizer(optimizer, args)              # the iteration the checkpoint was written at
        if start_i == 0:                                            # a reference-written checkpoint has no optimizer file
            start_i = history["train"]["iter"][-1] if history["train"]["iter"] else 0
        for g in optimizer.param_groups:                            # the restored (possibly decayed) learning rates
            if g["name"] == "sound":
                args.lr_sound = g["lr"]
            elif g["name"] == "frame_features":
                args.lr_frame = g["lr"]
            elif g["name"] == "synthesizer":
                args.lr_synthesizer = g["lr"]
    if args.mode == "eval":
        evaluate(wrapper, loader_val, history, 0, args, True, device, world)
        evaluate(wrapper, loader_val, history, 0, args, False, device, world)
        print("Evaluation Done!")
        return history
    av = _Cycle(dataset.make_loader(args.av_list_train, args, "train", args.batch_size, True, workers=args.workers))
    ao = _Cycle(dataset.make_loader(args.ao_list_train, args, "train", args.batch_size, True, workers=args.workers, seed=10))

    err_total = err_av = err_ao = match_sum = 0.0
    av_count = ao_count = 0
    t_iter = t_data = 0.0
    for i in range(start_i + 1, args.num_iters):
        tic = time.perf_counter()
        use_vis = av_ao_schedule(i, args)
        batch = dataset.to_device((av if use_vis else ao).next(), device)
        t_data += time.perf_counter() - tic
        if three_stage:
            err, match_loss = sopp.train_step_3stage(wrapper, batch, optimizer, use_vis, i, args)
        else:
            err, match_loss = train_step(wrapper, batch, optimizer, use_vis, args)  # one host sync, like the reference
        t_iter += time.perf_counter() - tic
        err_total += err
        if use_vis:
            err_av += err - match_loss * args.match_weight
            match_sum += match_loss
            av_count += 1
        else:
            err_ao += err
            ao_count += 1
        if i % args.disp_iter == 0 and i != 0:
            if rank == 0:
                print("iter: [{}/{}], Time: {:.2f}, Data: {:.2f}, lr_sound: {}, lr_frame: {}, loss: {:.3f}, loss_ao: {:.3f}, "
                      "loss_av: {:.3f} loss_match {:.3f}".format(
                          i, args.num_iters, t_iter / args.disp_iter, t_data / args.disp_iter, args.lr_sound, args.lr_frame,
                          err_total / args.disp_iter, err_ao / ao_count if ao_count else 0.66,
                          err_av / av_count if av_count else 0.25, match_sum / av_count if av_count else 0))
            history["train"]["iter"].append(i)
            history["train"]["err"].append(err_total / args.disp_iter)
            if ao_count:
                history["train_ao"]["iter"].append(i)
                history["train_ao"]["err"].append(err_ao / ao_count)
            if av_count:
                history["train_av"]["iter"].append(i)
                history["train_av"]["err"].append(err_av / av_count)
            err_total = err_av = err_ao = match_sum = 0.0
            av_count = ao_count = 0
            t_iter = t_data = 0.0
        # main.py:756-760 drops the learning rates AFTER writing the checkpoint of the same iteration; with the shipped flags
        # every lr step falls on a checkpoint iteration, so a checkpoint would hold the undecayed rates and a run resumed
        # from it (which starts at i + 1) would never decay.  Dropping first changes nothing else: the rates are next used
        # by step i + 1 either way.
        if i in args.lr_steps:
            adjust_learning_rate(optimizer, args)
        if i % args.eval_iter == 0 and i > 1:
            evaluate(wrapper, loader_val, history, i, args, True, device, world)
            evaluate(wrapper, loader_val, history, i, args, False, device, world)
            if rank == 0:
                ckpt.checkpoint(nets, history, i, args, optimizer=optimizer)
    print("Training Done!")
    return history


def cli(argv=None):
    args = ArgParser().parse_train_arguments(argv)
    print("Model ID: {}".format(args.id))
    args.ckpt = os.path.join(args.ckpt, args.id)                   # main.py:776-791
    three_stage = getattr(args, "train_steps", None) is not None
    paths = None
    if args.mode == "train":
        os.makedirs(args.ckpt, exist_ok=True)
        if args.load_ckpt:
            paths = ckpt.resume_paths(args, three_stage=three_stage)
    elif args.mode == "eval":
        paths = ckpt.resume_paths(args, best=True, three_stage=three_stage)
    if paths is not None:
        args.weights_sound, args.weights_frame = paths[:2]
        if three_stage:
            args.weights_synthesizer, args.weights_net_pit = paths[2:]
    args.best_err = float("inf")
    return main(args)


if __name__ == "__main__":
    cli()
