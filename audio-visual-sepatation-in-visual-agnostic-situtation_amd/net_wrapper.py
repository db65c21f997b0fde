"""Step definition: NetWrapper / create_optimizer / train_step — the host-side mirror of the
reference's main.py:39-192 and :536-569, on the HIP path.

``NetWrapper(nets, crit_ao, crit_av).forward(batch_data, args, use_vis, is_share=False)``
returns ``(err, outputs)`` with the reference's dictionary keys.  The batch contract is the
reference loader's (dataset/music.py:275-331): ``mag_mix [B,1,512,256]``, ``mags`` list of N,
``frames`` list of N ``[B,3,T,224,224]``; alternatively ``audios``/``audio_mix`` waveforms can be
given and the STFT runs on the GPU (``attach_stft``).
"""
import os

import torch

from . import kernels as K
from . import lib
from .lib import ACT_BY_NAME
from .models import activate
from .models.criterion import PitWrapper, mask_loss, pit_select

args = None  # module-global like main.py's `args`, used by train_step when none is passed


class NetWrapper(torch.nn.Module):
    def __init__(self, nets, crit_ao, crit_av):
        super().__init__()
        if len(nets) != 2:
            raise NotImplementedError("the 3-net form (net_motion, --load_clips) is out of scope")
        self.net_sound, self.net_frame = nets
        self.load_clips = False
        self.crit_ao = crit_ao
        self.crit_av = crit_av
        self.stft_plan = None

    # ------------------------------------------------------------------ main.py:51-95
    def prepare(self, batch_data, args, use_vis=True, is_share=False):
        mag_mix = batch_data["mag_mix"]
        lib.require_gpu(mag_mix)
        N = args.num_mix
        mags_in = torch.stack([m.float() for m in batch_data["mags"][:N]], 0).contiguous()
        mix_w, mags_w, log_mag_mix, weights, gt = K.prepare(
            mag_mix.float().contiguous(), mags_in, args.log_freq, args.weighted_loss, args.binary_mask)
        mags = [mags_w[n] for n in range(N)]
        for n in range(N):                       # the reference replaces the entries in place (:66)
            batch_data["mags"][n] = mags[n]
        gt_masks = [gt[n] for n in range(N)]
        self._gt_stack = gt                      # [N,B,1,F,T], one buffer for the fused loss
        if use_vis or args.fusion_type == "share" or is_share:
            return batch_data["frames"], None, mags, mix_w, log_mag_mix, gt_masks, weights
        return mags, mix_w, log_mag_mix, gt_masks, weights

    # ------------------------------------------------------------------ main.py:97-111
    def forward_ao(self, data, args):
        mags, mag_mix, log_mag_mix, gt_masks, weight = data
        self.unet_nodes = 1          # autograd nodes owning the U-Net parameters in this step (FlatSGD.arm_early_reduce)
        feat_sound, *_ = self.net_sound(log_mag_mix, None)
        act = ACT_BY_NAME.get(args.output_activation)
        if act is None:
            raise Exception("Unkown activation!")
        B = feat_sound.shape[0]
        pred, sums, FT = mask_loss(feat_sound, self._gt_stack, weight, act, "bce")   # PIT always uses BCE
        mat = sums / FT                                                              # [B,2,2] target x prediction
        loss, perms = pit_select(mat)                    # winning permutation picked on the device (no host sync)
        self._last_perms = perms
        err = loss.mean().to(torch.float32)
        pred_last = pred.permute(0, 2, 3, 1)                                          # B x F x T x C
        ordered = PitWrapper.reorder_tensor(pred_last, perms)
        gt = torch.stack(gt_masks, dim=-1)[:, 0]
        S = len(gt_masks)        # main.py:103 hard-codes 2; more sources: one weight copy per target
        w2 = torch.stack([weight[:, 0]] * S, dim=-1)
        return err, {"pred_masks": [ordered[..., i].unsqueeze(1) for i in range(S)],
                     "gt_masks": [gt[..., i].unsqueeze(1) for i in range(S)],
                     "mag_mix": mag_mix, "mags": mags, "weight": w2}

    # ------------------------------------------------------------------ main.py:117-121
    fork_sources = os.environ.get("AVSEP_FORK_SOURCES", "1") != "0"
    early_trunk = os.environ.get("AVSEP_EARLY_TRUNK", "1") != "0"

    def _frame_features(self, frames, N, args, early=False):
        """The visual trunk over each source's frames.  The passes are independent (the reference calls net_frame once per
        source), and a single pass leaves the chip partly idle in its tail rounds and small layers: the passes are issued on
        HIP streams of their own (forward here; autograd runs a node's backward on the stream of its forward), what they
        share is ordered by events (kernels.fork_streams, FlatSGD.node_finished).
        early=False: source 0 on the current stream, the others on side streams, joined before returning -> list of features.
        early=True (NetWrapper.forward, before the STFT): every source on a side stream; returns (features, join) and the
        caller joins when the features are needed — the STFT, the mask preparation and the U-Net encoder of the current stream
        run meanwhile."""
        def one(n):
            return activate(self.net_frame.forward_multiframe(frames[n], pool=args.not_pool_vis), args.img_activation)
        if not (self.fork_sources and N > 1 and frames[0].is_cuda):
            feats = [one(n) for n in range(N)]
            return (feats, None) if early else feats
        main = torch.cuda.current_stream()
        side = self.__dict__.setdefault("_src_streams", [])
        while len(side) < N:
            side.append(torch.cuda.Stream())
        fork = torch.cuda.Event()
        fork.record(main)                                   # everything the passes read (frames, updated weights) is older
        feats, used = [None] * N, []
        with K.fork_streams():
            for n in range(N):
                if n == 0 and not early:
                    feats[0] = one(0)
                    continue
                s = side[n]
                s.wait_event(fork)
                with torch.cuda.stream(s):
                    feats[n] = one(n)
                feats[n].record_stream(main)
                used.append(s)

        def join():
            for s in used:
                main.wait_stream(s)
        if early:
            return feats, join
        join()
        return feats

    # ------------------------------------------------------------------ main.py:113-148
    def forward_av(self, data, args, early_feats=None):
        N = args.num_mix
        frames, _, mags, mag_mix, log_mag_mix, gt_masks, weight = data
        join = None
        if early_feats is not None:
            feat_frames, join = early_feats
        else:
            feat_frames = self._frame_features(frames, N, args)
        kind = getattr(self.crit_av, "kind", "bce")
        act = ACT_BY_NAME.get(args.output_activation)
        if act is None:
            raise Exception("Unkown activation!")
        fused = args.output_activation != "softmax"   # softmax over the singleton channel is identically 1 (:132)
        gt_nat = self._gt_stack
        errs, sums_both, matches = [], [], []
        # pass 1: visual features in reversed order against reversed targets; pass 2: natural order.
        # Both passes read the same spectrogram: the U-Net shares its encoder between them (forward_pair).
        pair = hasattr(self.net_sound, "forward_pair") and getattr(self, "share_encoder", True) and 2 <= N <= 4
        if join is not None and not (pair and all(t.is_contiguous() and t.dtype == torch.float32 for t in feat_frames)):
            join()                                     # no node that could take the join between its encoder and its decoder
            join = None
        if pair:
            # `before_decode`: the trunk's streams are joined after the encoder has been issued (it does not read the features)
            passes = self.net_sound.forward_pair(log_mag_mix, feat_frames[::-1], feat_frames, before_decode=join)
            self.unet_nodes = 1 if getattr(self.net_sound, "extra_size", None) is None else 2
        else:
            passes = None
            self.unet_nodes = 2
        for pi, reverse in enumerate((True, False)):
            vis_in = feat_frames[::-1] if reverse else feat_frames
            feat_sound, meta = passes[pi] if passes is not None else self.net_sound(log_mag_mix, vis_in)
            if fused:
                gt = gt_nat.flip(0).contiguous() if reverse else gt_nat
                pred, sums, FT = mask_loss(feat_sound, gt, weight, act, kind)
                B, S = feat_sound.shape[:2]
                sums_both.append(sums)
                pred_masks = [pred[:, n].unsqueeze(1) for n in range(N)]
            else:
                pred_masks = [activate(feat_sound[:, n].unsqueeze(1), args.output_activation) for n in range(N)]
                errs.append(self.crit_av(pred_masks, gt_masks[::-1] if reverse else gt_masks, weight).reshape(1))
            matches.append(meta[0])
        match_loss = matches[0] + matches[1]
        if fused:
            # (err_rev + err_nat) / 2 with err = trace-sum / (B S FT): one reduction over both passes' [B,S,S] sums
            tot = torch.diagonal(torch.stack(sums_both), dim1=2, dim2=3).sum()
            mean_err = (tot * (0.5 / (B * S * FT))).to(torch.float32)
        else:
            mean_err = (errs[0] + errs[1]) / 2
        err = torch.add(mean_err, match_loss, alpha=args.match_weight).reshape(1)
        return err, {"pred_masks": pred_masks, "gt_masks": gt_masks, "mag_mix": mag_mix, "mags": mags,
                     "weight": weight, "match_loss": match_loss.reshape(1), "att_maps": meta[1],
                     "logits": feat_sound}

    # ------------------------------------------------------------------ main.py:162-192 (MixVis)
    def forward_avmiximg(self, data, args):
        frames, _, mags, mag_mix, log_mag_mix, gt_masks, weight = data
        mix_frame = torch.cat(list(frames), dim=-1)                          # B x 3 x T x H x (W*S)
        feat_frame = activate(self.net_frame.forward_multiframe(mix_frame, pool=args.not_pool_vis),
                              args.img_activation)
        self.unet_nodes = 1
        feat_sound, meta = self.net_sound(log_mag_mix, [feat_frame])
        act = ACT_BY_NAME.get(args.output_activation)
        if act is None:
            raise Exception("Unkown activation!")
        pred, sums, FT = mask_loss(feat_sound, self._gt_stack, weight, act, "bce")
        mat = sums / FT
        loss, perms = pit_select(mat)
        err = loss.mean().to(torch.float32).reshape(1)
        ordered = PitWrapper.reorder_tensor(pred.permute(0, 2, 3, 1), perms)
        gt = torch.stack(gt_masks, dim=-1)[:, 0]
        match_loss = meta[0].reshape(1)
        err = err + match_loss * args.match_weight
        return err, {"pred_masks": [ordered[..., i].unsqueeze(1) for i in range(2)],
                     "gt_masks": [gt[..., i].unsqueeze(1) for i in range(2)],
                     "mag_mix": mag_mix, "mags": mags, "weight": weight, "match_loss": match_loss, "maps": meta[1]}

    # ------------------------------------------------------------------ main.py:150-160
    def forward(self, batch_data, args, use_vis, is_share=False):
        sink = getattr(self.net_sound, "_grad_sink", None)
        if sink is not None and torch.is_grad_enabled():
            sink.note_caller_stream()          # the stream a following .backward() is ordered on (FlatSGD._end_of_backward)
        early = None
        # Under HIP-graph capture source 0 stays on the capturing stream (early=False): with EVERY trunk pass on a side stream
        # hipStreamEndCapture crashed on ROCm 7.0; the topology below is the one that captures and replays.
        capturing = batch_data["frames"][0].is_cuda and torch.cuda.is_current_stream_capturing() if use_vis else False
        if use_vis and args.fusion_type != "MixVis" and self.fork_sources and self.early_trunk and args.num_mix > 1 and \
                batch_data["frames"][0].is_cuda and not capturing:
            # the visual trunk does not read the spectrograms: issue its passes first, on their streams
            early = self._frame_features(batch_data["frames"], args.num_mix, args, early=True)
        if "mag_mix" not in batch_data:
            self.attach_stft(batch_data, args)
        data = self.prepare(batch_data, args, use_vis, is_share)
        if use_vis:
            if args.fusion_type == "MixVis":
                return self.forward_avmiximg(data, args)
            return self.forward_av(data, args, early_feats=early)
        return self.forward_ao(data, args)

    # ------------------------------------------------------------------ dataset/base.py:142-147,174-189 on the GPU
    def attach_stft(self, batch_data, args):
        """mag_mix / mags / phase_mix from the waveforms in the batch (N+1 STFTs per mixture)."""
        wavs = [batch_data["audio_mix"]] + list(batch_data["audios"][:args.num_mix])
        B, Ln = wavs[0].shape
        if self.stft_plan is None:
            self.stft_plan = K.Stft(wavs[0].device, args.stft_frame, args.stft_hop,
                                    getattr(args, "stft_pad_mode", "reflect"))
        rows = torch.cat([w.float() for w in wavs], 0).contiguous()             # [(1+N)*B, L]
        mag, phase = self.stft_plan.stft(rows, want_phase=True)
        mag = mag.view(len(wavs), B, 1, *mag.shape[1:])
        batch_data["mag_mix"] = mag[0]
        batch_data["mags"] = [mag[1 + n] for n in range(args.num_mix)]
        batch_data["phase_mix"] = phase.view(len(wavs), B, 1, *phase.shape[1:])[0]
        return batch_data


# ---------------------------------------------------------------------- main.py:536-555
class FlatSGD:
    """torch.optim.SGD semantics (momentum, weight decay, per-group lr; main.py:547) over flat
    buffers: every parameter becomes a view into one fp32 parameter buffer and its .grad a view
    into one gradient buffer, so zero_grad is one memset, the data-parallel all-reduce is ONE
    RCCL call over xGMI (replacing DataParallel's broadcast + reduce, main.py:661) and the
    update is one fused HIP launch per group."""

    ALIGN = 64      # elements: 256 bytes

    def __init__(self, groups, momentum=0.9, weight_decay=0.0, process_group=None, world_size=1, overlap=None,
                 require_gpu=True, force_collective=False):
        """`force_collective`: issue the early + late all-reduce even at world size 1 (a 1-rank RCCL group on a one-GPU
        box: the collectives, their stream order against the step's kernels and the hooks run exactly as on N ranks).
        `require_gpu=False` only skips the device check so that the bucket / all-reduce logic (`reduce_gradients`) can be
        exercised on CPU tensors over gloo (tests/test_dp_gloo.py); `step()` itself has no CPU path."""
        self.momentum, self.weight_decay = momentum, weight_decay
        self.world_size, self.process_group = world_size, process_group
        self._dist = world_size > 1 or bool(force_collective)
        # Data parallel: the first group's gradients (the U-Net: 130 MB of the 180 MB) are complete as soon as its
        # autograd node has run, i.e. BEFORE the visual trunk's backward (~30 ms) starts: their all-reduce is issued
        # right there (asynchronously, RCCL's own stream) and overlaps that backward; step() reduces the rest.
        self.overlap = (os.environ.get("AVSEP_DP_OVERLAP", "1") != "0") if overlap is None else overlap
        self._early, self.early_reductions = None, 0
        self._pending, self._nodes_left, self._armed = set(), 0, False
        self._slot, self._written, self._returned = {}, set(), set()       # direct gradient placement (grad_dest)
        self._scratch = None
        self._scratch_src = {}
        self._node_events = []
        self._end_queued, self._at_end = False, []
        self._caller_stream = None
        self._auto_step = None          # (only,) while a step is to be issued by the end-of-backward callback; "done" after it ran
        self.param_groups = []
        params = []
        for g in groups:
            ps = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append({"params": ps, "lr": g["lr"], "name": g.get("name", str(len(self.param_groups)))})
            params += ps
        if not params:
            raise ValueError("no parameters")
        self._reports = False          # True once attach() hands this optimizer to networks whose nodes call node_finished
        dev = params[0].device
        if require_gpu:
            lib.require_gpu(params[0])
        # every tensor starts on a 256-byte boundary of the flat buffers (the wide loads / stores of the kernels that
        # write gradients in place, the SGD kernel and RCCL all see aligned rows); the padding holds zeros throughout
        A = self.ALIGN
        total = sum((p.numel() + A - 1) // A * A for p in params)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_buf = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._views = []
        self._group_of = {}
        for gi, g in enumerate(self.param_groups):
            g["range"] = [off, off]
            for p in g["params"]:
                self._group_of[p] = gi
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p.data)
                gv = self.flat_grad[off:off + n].view_as(p.data)
                p.grad = gv
                self._views.append((p, gv))
                self._slot[p] = gv
                off += (n + A - 1) // A * A
            g["range"][1] = off
        if self._dist and self.overlap:
            for p in self.param_groups[0]["params"]:
                p.register_post_accumulate_grad_hook(self._on_first_group_grad)

    # ---- direct gradient placement: the backward kernels write into the flat buffer ------------------------------------
    def grad_dest(self, p):
        """The zeroed flat .grad view of `p` if a kernel may WRITE this step's first contribution into it, else None
        (a second contribution, a detached .grad, a channels-last view): the caller then returns the gradient to
        autograd, whose AccumulateGrad adds it into the view."""
        gv = self._slot.get(p)
        if gv is None or p in self._written or p.grad is None or p.grad.data_ptr() != gv.data_ptr():
            return None
        self._written.add(p)
        return gv

    def scratch_dest(self, p):
        """For a parameter another node of this step has already placed: a view, at the same offset, of a second flat
        buffer that the node's kernels may write; fold_scratch() then adds all of them into the gradient buffer with ONE
        launch per contiguous run (the visual trunk runs once per source: its second node used to cost one
        AccumulateGrad add per parameter)."""
        gv = self._slot.get(p)
        if gv is None or p not in self._written or p.grad is None or p.grad.data_ptr() != gv.data_ptr():
            return None
        gi = self._group_of[p]
        sc = self._scratch_of_current_stream(gi, create=True)
        self._scratch_src[p] = sc              # fold_scratch adds THIS buffer, whichever stream folds
        off = gv.storage_offset() - self.param_groups[gi]["range"][0]
        return sc[off:off + gv.numel()].view_as(gv)

    def _scratch_of_current_stream(self, gi, create=False):
        """One scratch buffer per (stream, parameter group), covering that group's flat range only: nodes that run on different
        streams (the visual trunk's passes, fork_streams) must not overwrite one another's not yet folded contributions, and
        a stream only ever holds second contributions of the network its nodes belong to (a trunk stream never needs the
        U-Net's 130 MB)."""
        if self._scratch is None:
            self._scratch = {}
        key = (torch.cuda.current_stream(self.flat_grad.device) if self.flat_grad.is_cuda else None, gi)
        sc = self._scratch.get(key)
        if sc is None and create:
            a, b = self.param_groups[gi]["range"]
            sc = self._scratch[key] = torch.zeros(b - a, dtype=self.flat_grad.dtype, device=self.flat_grad.device)
        return sc

    def _wait_for_nodes(self):
        """Order the current stream behind every autograd node of this step that has reported (node_finished): their kernels
        wrote into the flat gradient buffer on whatever stream their forward ran on."""
        if self._node_events:
            cur = torch.cuda.current_stream(self.flat_grad.device)
            for ev, st in self._node_events:
                if st != cur:
                    cur.wait_event(ev)

    def fold_scratch(self, params):
        """flat_grad += scratch over the flat ranges of `params` (merged into contiguous runs; padding holds zeros)."""
        A = self.ALIGN
        by_buf = {}
        for p in params:
            sc = self._scratch_src.get(p)
            if sc is None:
                raise lib.AvsepError("fold_scratch: a parameter that was never handed a scratch view (scratch_dest) in this step")
            by_buf.setdefault(id(sc), (sc, self._group_of[p], []))[2].append(p)
        self._wait_for_nodes()                      # the first contributions (and earlier folds) may be on other streams
        for sc, gi, ps in by_buf.values():          # one scratch buffer = one (stream, group): runs never span groups
            base = self.param_groups[gi]["range"][0]
            spans = sorted((self._slot[p].storage_offset(), (self._slot[p].numel() + A - 1) // A * A) for p in ps)
            runs = []
            for off, n in spans:
                if runs and runs[-1][1] == off:
                    runs[-1][1] = off + n
                else:
                    runs.append([off, off + n])
            for a, b in runs:
                self.flat_grad[a:b].add_(sc[a - base:b - base])
            for p in ps:
                del self._scratch_src[p]

    def node_finished(self, group, returned):
        """An autograd node of network `group` has run its backward; `returned` = the parameters whose gradient it handed
        back to autograd (their AccumulateGrad — and post-accumulate hook — is still to come), the others are in place."""
        self._returned.update(returned)
        if self.flat_grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._node_events.append((ev, torch.cuda.current_stream(self.flat_grad.device)))
            self._queue_end_of_backward()
        if group != self.param_groups[0]["name"] or not self._armed:
            return
        self._nodes_left -= 1
        if self._nodes_left == 0:
            self._pending -= (self._written - self._returned)       # placed by a kernel and complete: no hook will fire
            self._maybe_start_early_reduce()

    def _queue_end_of_backward(self):
        """Nodes whose forward ran on a forked stream wrote their gradients on that stream and handed autograd nothing to
        synchronise on: when the engine has run the last node of this backward pass, order the CALLER's stream behind every
        node (the contract of .backward(): its results are usable on the calling stream) and run the deferred releases."""
        if not self._end_queued:
            self._end_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def at_end_of_backward(self, fn):
        self._at_end.append(fn)
        self._queue_end_of_backward()

    def note_caller_stream(self):
        """Remember the stream the step is being issued on: called on the CALLER's thread by NetWrapper.forward (every forward
        pass that can be followed by a backward pass) and by arm_early_reduce (between forward and backward) — not by
        zero_grad(), which a caller may run on another stream or skip between two accumulated backward passes."""
        self._caller_stream = torch.cuda.current_stream(self.flat_grad.device) if self.flat_grad.is_cuda else None

    def _end_of_backward(self):
        # The engine may run this on one of its worker threads, whose current stream is that thread's default: the stream to
        # order is the CALLER's — the one the forward pass of this backward was issued on (note_caller_stream).
        self._end_queued = False
        caller = self._caller_stream
        if caller is None or not self.flat_grad.is_cuda:
            self._wait_for_nodes()
            fns, self._at_end = self._at_end, []
            for fn in fns:
                fn()
            return
        with torch.cuda.stream(caller):
            self._wait_for_nodes()
            fns, self._at_end = self._at_end, []
            for fn in fns:
                fn()
            if isinstance(self._auto_step, tuple):
                # The fused SGD launches go out HERE, from the engine's last callback, instead of after the host has come back
                # from .backward() through the engine's thread hand-off (0.6-0.8 ms of idle GPU in front of sgd_kernel, measured
                # on the bf16 step); same stream, same order as an explicit step() right after backward().
                only, self._auto_step = self._auto_step[0], None
                self.step(only)
                self._auto_step = "done"

    def step_after_backward(self, only=None):
        """Ask the end-of-backward callback to issue step(only).  Single-process only (the data-parallel step waits for its
        collectives on the caller's thread); returns False when nothing was armed.  train_step_async then asks
        took_auto_step(): True = the update is already in flight on the caller's stream, False = call step() as usual (no
        node of this backward pass reported, so no callback ran)."""
        if self._dist or not self._reports or not self.flat_grad.is_cuda:
            return False
        self._auto_step = (only,)
        return True

    def took_auto_step(self):
        done, self._auto_step = self._auto_step == "done", None
        return done

    def _on_first_group_grad(self, p):
        """Fires once per parameter of the first group after autograd accumulated into its flat view."""
        if self._armed:
            self._pending.discard(p)
            self._maybe_start_early_reduce()

    def _maybe_start_early_reduce(self):
        if self._armed and not self._pending and self._nodes_left <= 0 and self._early is None and self._flat_views_intact():
            import torch.distributed as dist
            self._wait_for_nodes()
            a, b = self.param_groups[0]["range"]
            self._early = dist.all_reduce(self.flat_grad[a:b], group=self.process_group, async_op=True)
            self.early_reductions += 1
            self._armed = False

    def _flat_views_intact(self):
        g = self.param_groups[0]
        return all(p.grad is not None and p.grad.data_ptr() == gv.data_ptr()
                   for p, gv in self._views[:len(g["params"])])

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, gv in self._views:
            p.grad = gv
        self._written.clear()
        self._returned.clear()
        self._scratch_src = {}
        self._node_events = []
        self._end_queued, self._at_end = False, []       # (a backward pass that raised never ran its final callback)
        self._early, self._armed, self._nodes_left = None, False, 0     # disarmed until arm_early_reduce()
        self._pending = set()

    def arm_early_reduce(self, accumulations=1):
        """Call between forward and backward; `accumulations` = autograd nodes of the first group's network in this step's
        graph.  A first-group parameter is complete when every such node has run AND every gradient a node handed back to
        autograd has been accumulated (AccumulateGrad runs once per leaf and backward, then the post-accumulate hook);
        parameters whose gradient the kernels placed directly (grad_dest) and no node returned need no hook.  Nodes that
        do not report (plain torch modules): the hooks alone count, as before."""
        self.note_caller_stream()
        if self._dist and self.overlap and accumulations > 0:
            self._armed = True
            self._pending = set(self.param_groups[0]["params"])
            self._nodes_left = accumulations if self._reports else 0

    def _collect(self):
        # a caller that ran module.zero_grad(set_to_none=True) (torch default, main.py:560) left
        # autograd to allocate fresh .grad tensors: fold them back into the flat buffer
        for g in self.param_groups:
            g["no_grad"] = bool(g["params"]) and all(p.grad is None for p in g["params"])
        for p, gv in self._views:
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv

    def _offsets(self, align):
        """Start of every parameter inside a flat buffer whose tensors are padded to `align` elements."""
        offs, off = [], 0
        for g in self.param_groups:
            for p in g["params"]:
                offs.append(off)
                off += (p.numel() + align - 1) // align * align
        return offs, off

    def state_dict(self):
        """What torch.optim.SGD.state_dict carries, independent of this class's buffer layout: one momentum tensor per
        parameter in its logical shape (group order, then parameter order), plus per-group lr / started flags."""
        offs, _ = self._offsets(self.ALIGN)
        params = [p for g in self.param_groups for p in g["params"]]
        buf = self.flat_buf.detach()
        return {"format": 2,
                "momentum": [buf[o:o + p.numel()].view(p.shape).cpu().clone() for o, p in zip(offs, params)],
                "groups": [{"name": g["name"], "lr": g["lr"], "started": bool(g.get("started", False)),
                            "numel": sum(p.numel() for p in g["params"])} for g in self.param_groups]}

    def load_state_dict(self, state):
        """Accepts the per-parameter format above and the two flat layouts earlier builds wrote (`momentum_buffer` with
        group `range`s: unpadded, or padded to 64 elements).  Returns True when the momentum was restored; on a blob
        that matches neither (another architecture, another ALIGN) it warns, keeps zero momentum and restores only the
        per-group lr / started flags where the group names agree — a history-only resume instead of a crash."""
        import warnings
        params = [p for g in self.param_groups for p in g["params"]]
        offs, _ = self._offsets(self.ALIGN)
        ok = False
        if state.get("format") == 2:
            mom = state["momentum"]
            ok = len(mom) == len(params) and all(tuple(m.shape) == tuple(p.shape) for m, p in zip(mom, params))
            if ok:
                for o, p, m in zip(offs, params, mom):
                    self.flat_buf[o:o + p.numel()].copy_(m.reshape(-1).to(self.flat_buf.device))
        elif "momentum_buffer" in state:
            flat = state["momentum_buffer"]
            for align in (self.ALIGN, 1):
                src, total = self._offsets(align)
                ends, off = [], 0
                for g in self.param_groups:
                    beg = off
                    off += sum((p.numel() + align - 1) // align * align for p in g["params"])
                    ends.append([beg, off])
                if flat.numel() == total and [list(g["range"]) for g in state["groups"]] == ends:
                    for o, so, p in zip(offs, src, params):
                        self.flat_buf[o:o + p.numel()].copy_(flat[so:so + p.numel()].to(self.flat_buf.device))
                    ok = True
                    break
        if not ok:
            warnings.warn("optimizer state does not match the parameter groups (written by another build or for another "
                          "architecture): momentum starts from zero, learning rates are restored by group name")
        by_name = {s["name"]: s for s in state.get("groups", [])}
        for g in self.param_groups:
            s = by_name.get(g["name"])
            if s is not None:
                g["lr"] = s["lr"]
                g["started"] = bool(s["started"]) and ok
        return ok

    def reduce_gradients(self, only=None):
        """The data-parallel half of step(): fold stray .grad tensors back into the flat buffer, wait for the early
        all-reduce of the first group (if it was issued) and sum the rest of the active range over the ranks with ONE
        all-reduce.  Returns (active groups, scale): flat_grad[range] * scale is the mean gradient over the ranks."""
        self._wait_for_nodes()
        self._collect()
        active = [g for g in self.param_groups
                  if (only is None or g["name"] in only) and g["range"][1] > g["range"][0] and not g.get("no_grad")]
        scale = 1.0
        if self._dist and active:
            import torch.distributed as dist
            rest = active
            if self._early is not None:                     # the first group is already being reduced
                self._early.wait()
                self._early = None
                rest = [g for g in active if g is not self.param_groups[0]]
            if rest:
                lo, hi = min(g["range"][0] for g in rest), max(g["range"][1] for g in rest)
                dist.all_reduce(self.flat_grad[lo:hi], group=self.process_group)   # ONE RCCL sum over xGMI
            scale = 1.0 / self.world_size
        return active, scale

    def step(self, only=None):
        """`only`: names of the groups that took part in this step's graph.  torch.optim.SGD skips a
        parameter whose .grad is None (no weight decay, no momentum update); with the reference's
        model.zero_grad() (set_to_none) that is every net_frame parameter on an audio-only step."""
        active, scale = self.reduce_gradients(only)
        for g in active:
            a, b = g["range"]
            first = not g.get("started", False)
            K.sgd_momentum_(self.flat_param[a:b], self.flat_grad[a:b], self.flat_buf[a:b], g["lr"],
                            self.momentum, self.weight_decay, scale, first)
            g["started"] = True


def attach_grad_sink(opt, *nets):
    """Let the autograd nodes of `nets` write parameter gradients straight into `opt`'s flat gradient buffer."""
    for n in nets:
        object.__setattr__(n, "_grad_sink", opt)
    opt._reports = True
    return opt


def create_optimizer(nets, args, process_group=None, world_size=1, force_collective=False):
    (net_sound, net_frame) = nets
    groups = [{"params": list(net_sound.parameters()), "lr": args.lr_sound, "name": "sound"},
              {"params": list(net_frame.fc.parameters()), "lr": args.lr_sound, "name": "frame_fc"}]
    if not args.fix_vis:
        groups.append({"params": list(net_frame.features.parameters()), "lr": args.lr_frame,
                       "name": "frame_features"})
    return attach_grad_sink(FlatSGD(groups, momentum=args.beta1, weight_decay=args.weight_decay,
                                    process_group=process_group, world_size=world_size,
                                    force_collective=force_collective), net_sound, net_frame)


def adjust_learning_rate(optimizer, args):
    args.lr_sound *= 0.1
    args.lr_frame *= 0.1
    if hasattr(args, "lr_motion"):
        args.lr_motion *= 0.1
    if hasattr(args, "lr_synthesizer"):          # SoP++/main.py:649-654
        args.lr_synthesizer *= 0.1
    for param_group in optimizer.param_groups:
        param_group["lr"] *= 0.1


def train_step_async(model, batch, optimizer, use_vis, step_args=None):
    """One train step without any host synchronisation: returns device tensors (err, match_loss, outputs)."""
    a = step_args if step_args is not None else args
    torch.set_grad_enabled(True)
    model.train()
    optimizer.zero_grad()
    with K.pack_scope():                 # the weights are constant from here to optimizer.step(): pack each image once
        err, outputs = model.forward(batch, a, use_vis)
        err = err.mean()
        only = None if use_vis else ("sound",)     # the visual net is not in an audio-only graph
        if isinstance(optimizer, FlatSGD):
            optimizer.arm_early_reduce(getattr(model, "unet_nodes", 0))
            optimizer.step_after_backward(only)
        err.backward()
    if isinstance(optimizer, FlatSGD):
        if not optimizer.took_auto_step():
            optimizer.step(only=only)
    else:
        optimizer.step()
    match_loss = outputs["match_loss"].mean() if use_vis else None
    return err.detach(), (match_loss.detach() if match_loss is not None else None), outputs


def train_step(model, batch, optimizer, use_vis, step_args=None):
    """main.py:557-569: zero_grad -> forward -> err.mean().backward() -> optimizer.step();
    returns (err.item(), match_loss.item() or None) like the reference (one host sync)."""
    err, match_loss, _ = train_step_async(model, batch, optimizer, use_vis, step_args)
    return err.item(), (match_loss.item() if match_loss is not None else None)
