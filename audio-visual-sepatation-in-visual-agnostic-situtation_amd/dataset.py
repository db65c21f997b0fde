"""Input-pipeline contract (reference: dataset/base.py:17-215, dataset/music.py:52-331,
dataset/video_transforms.py).

``MUSICMixDataset(csv_path, params, split=..., seed=..., random_sample=..., vis_data=...)`` reads the reference's
6-column lists (``audio.wav, frame_dir, num_frames, fps, audio_seconds, class``; rows with < 2 fields skipped),
repeats / shuffles them like the reference and draws a mixture per index with the SAME sequence of ``random``
calls (``random.seed(index)`` first), so the chosen clips, centre times, per-source gains and frame file names are
those the reference would draw:

* partner selection ``dc`` / ``sc`` / ``sv`` / ``random`` / ``vis1`` (music.py:56-92) and the chained
  ``random.random()`` strategy draw (music.py:283-289);
* centre time ~ U(start, end) with the margin rule, up to 10 tries against silence (music.py:96-130);
* window cut, U(0.5,1.5) gain in train, clip to +-1, divide by N, mixture = sum (base.py:156-172, music.py:120,127);
* frame indices ``round(t*fps) + (i - T//2)*stride`` or the ``one_frame`` random shift (music.py:132-156);
* frames: bicubic resize (shorter side to 1.1*imgSize in train, imgSize in val; long side capped at 448),
  random/centre crop, random flip (train), /255, ImageNet mean/std, stacked to [3,T,H,W] (base.py:96-110).

What is different, on purpose: the loader does NOT run the STFT (base.py:142-147 does it with librosa on the CPU
workers) — batches carry waveforms and ``NetWrapper.attach_stft`` computes mag/phase on the GPU; ``collate``
stacks into pinned host memory for asynchronous H2D copies.  Decoding uses ``scipy.io.wavfile`` (+ polyphase
resampling when the file's rate differs from audRate; librosa's resampler is not in this image) and PIL.
"""
import csv
import os
import random

import numpy as np
import torch

MUSIC11_CLASSES = ["accordion", "acoustic_guitar", "cello", "clarinet", "erhu", "flute", "saxophone", "trumpet",
                   "tuba", "violin", "xylophone"]
_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def read_list(csv_path):
    with open(csv_path, "r") as f:
        return [row for row in csv.reader(f, delimiter=",") if len(row) >= 2]


# ------------------------------------------------------------------------------------------ decoding
def read_wav_segment(path, rate, offset, duration):
    """Mono float32 samples of [offset, offset+duration) seconds at `rate` (librosa.load(sr, mono, offset, duration))."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path, mmap=True)
    a, b = int(round(max(offset, 0.0) * sr)), int(round((max(offset, 0.0) + duration) * sr))
    seg = np.asarray(data[a:b])
    if seg.dtype.kind == "i":
        seg = seg.astype(np.float32) / float(np.iinfo(seg.dtype).max + 1)
    elif seg.dtype.kind == "u":
        seg = (seg.astype(np.float32) - 128.0) / 128.0
    else:
        seg = seg.astype(np.float32)
    if seg.ndim == 2:
        seg = seg.mean(axis=1)
    if sr != rate and len(seg):
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(rate))
        seg = resample_poly(seg, rate // g, sr // g).astype(np.float32)
    return seg, rate


def _resize_size(w, h, size, max_size):
    # torchvision.transforms.functional.resize with an int size: shorter side -> size, long side capped at max_size
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_short, new_long) if w <= h else (new_long, new_short)


class VideoTransform:
    """The reference's Compose of video_transforms (one crop / flip decision per clip, shared by its frames)."""

    def __init__(self, img_size, train, max_size=448):
        self.img_size, self.train, self.max_size = img_size, train, max_size

    def __call__(self, frames):
        from PIL import Image
        size = int(self.img_size * 1.1) if self.train else self.img_size
        frames = [f.resize(_resize_size(f.size[0], f.size[1], size, self.max_size), Image.BICUBIC) for f in frames]
        th = tw = self.img_size
        w, h = frames[0].size
        if self.train:
            if w == tw and h == th:
                i = j = 0
            else:
                i = random.randint(0, h - th)
                j = random.randint(0, w - tw)
            frames = [f.crop((j, i, j + tw, i + th)) for f in frames]
            if random.random() < 0.5:
                frames = [f.transpose(Image.FLIP_LEFT_RIGHT) for f in frames]
        else:
            out = []
            for f in frames:
                w, h = f.size
                i, j = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
                out.append(f.crop((j, i, j + tw, i + th)))
            frames = out
        mean = torch.tensor(_MEAN).view(3, 1, 1)
        std = torch.tensor(_STD).view(3, 1, 1)
        ts = [(torch.from_numpy(np.asarray(f, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0) - mean) / std
              for f in frames]
        return torch.stack(ts, dim=1)                                   # [3,T,H,W]


# ------------------------------------------------------------------------------------------ dataset
class MUSICMixDataset(torch.utils.data.Dataset):
    def __init__(self, csv_path, params, split="val", debug=False, seed=None, random_sample=False, vis_data=False):
        p = params
        self.debug = debug
        self.num_frames, self.imgSize, self.stride_frames = p["num_frames"], p["imgSize"], p["stride_frames"]
        self.one_frame = p["one_frame"]
        if p.get("load_clips"):
            raise NotImplementedError("load_clips (mmaction clip pipeline for net_motion) is out of scope")
        self.audRate, self.audLen = p["audRate"], p["audLen"]
        self.audSec = 1.0 * self.audLen / self.audRate
        self.num_mix = p["num_mix"]
        self.classes = MUSIC11_CLASSES
        self.rate_dc, self.rate_sc, self.rate_sv = p["rate_dc"], p["rate_sc"], p["rate_sv"]
        self.margin, self.max_silent = p["margin"], p["max_silent"]
        self.class_int_map = {k: v for v, k in enumerate(self.classes)}
        self.split = split
        self.seed = seed if seed is not None else p["seed"]
        self.random_sample, self.vis_data = random_sample, vis_data
        random.seed(self.seed)
        self.vid_transform = VideoTransform(self.imgSize, split == "train")
        if isinstance(csv_path, str):
            self.list_samples = read_list(csv_path)
        elif isinstance(csv_path, list):
            self.list_samples = csv_path
        else:
            raise TypeError("Error list_samples!")
        print(f"Length of dataset: {len(self.list_samples)}")
        self.dict_samples = {}
        for s in self.list_samples:
            self.dict_samples.setdefault(s[-1], []).append(s)
        if split == "train":
            self.list_samples = self.list_samples * p["train_repeat"]
            if not debug:
                random.shuffle(self.list_samples)
        else:
            self.list_samples = self.list_samples * p["val_repeat"]
        assert len(self.list_samples) > 0

    def __len__(self):
        return len(self.list_samples)

    # ---- music.py:56-92
    def get_samples(self, index, option="dc"):
        infos = [self.list_samples[index]]
        sound_cls = self.list_samples[index][5]
        if option == "dc":
            left = self.classes.copy()
            for _ in range(self.num_mix - 1):
                left.remove(sound_cls)          # (sic) raises ValueError for num_mix > 2, like the reference
                infos.append(random.choice(self.dict_samples[random.choice(left)]))
        elif option == "sc":
            for _ in range(self.num_mix - 1):
                infos.append(random.choice(self.dict_samples[sound_cls]))
        elif option == "sv":
            infos += [self.list_samples[index]] * (self.num_mix - 1)
        elif option == "random":
            for _ in range(self.num_mix - 1):
                infos.append(self.list_samples[random.randint(0, len(self.list_samples) - 1)])
        elif option == "vis1":
            infos = [random.choice(self.dict_samples["cello"])]
            for _ in range(self.num_mix - 1):
                infos.append(random.choice(self.dict_samples["flute"]))
        assert self.num_mix == len(infos)
        return infos

    def choose(self, index):
        """The strategy draw of music.py:283-289 (each elif draws a fresh random number, like the reference)."""
        if self.random_sample:
            return self.get_samples(index, "random")
        if self.vis_data:
            return self.get_samples(index, self.vis_data)
        if random.random() < self.rate_dc:
            return self.get_samples(index, "dc")
        if random.random() < self.rate_dc + self.rate_sc:
            return self.get_samples(index, "sc")
        if random.random() < self.rate_dc + self.rate_sc + self.rate_sv:
            return self.get_samples(index, "sv")
        raise UnboundLocalError("no sampling strategy drawn (rate_dc + rate_sc + rate_sv < 1): the reference fails here too")

    # ---- base.py:149-172
    def _load_audio_file(self, path, center_t):
        assert path.endswith(".wav")
        return read_wav_segment(path, self.audRate, center_t - self.margin - self.audSec / 2, self.margin * 2 + self.audSec)

    def _load_audio(self, path, center_t):
        audio = np.zeros(self.audLen, dtype=np.float32)
        raw, _ = self._load_audio_file(path, center_t)
        center_idx = int((self.margin + self.audSec / 2) * self.audRate)
        start = max(0, center_idx - self.audLen // 2)
        end = min(len(raw), center_idx + self.audLen // 2 + self.audLen % 2)
        audio[:max(end - start, 0)] = raw[start:end]
        if self.split == "train":
            audio *= random.random() + 0.5
        np.clip(audio, -1.0, 1.0, out=audio)
        return audio

    # ---- music.py:96-130
    def get_audios(self, infos):
        audios, center_times = [], []
        for n in range(self.num_mix):
            apath, _, num_f, fps, a_len, _ = infos[n]
            act_len = min(int(num_f) / float(fps), float(a_len))
            for j in range(10):
                end = act_len - self.margin - self.audSec / 2
                start = self.margin + self.audSec / 2
                if start > end:
                    end = act_len - self.audSec / 2
                    start = self.audSec / 2
                t = random.uniform(0 + start, end)
                aud = self._load_audio(apath, t)
                if self.split == "train":
                    silent = bool(np.all(aud == 0))
                else:
                    silent = ((np.abs(aud) < 0.001).sum() / self.audLen) > self.max_silent
                if not silent or j == 9:
                    if silent:
                        print(f"Load {apath} failed.")
                    center_times.append(t)
                    audios.append(aud / self.num_mix)
                    break
        return audios, np.asarray(audios).sum(axis=0), center_times

    # ---- music.py:132-156
    def frame_paths(self, info, center_t):
        _, fpath, _, fps, _, _ = info
        center_idx = round(center_t * float(fps))
        if self.one_frame:
            shift = random.randint(int(-1 * self.stride_frames), int(1 * self.stride_frames))
            return [os.path.join(fpath, "{:06d}.jpg".format(center_idx + shift))], center_idx
        return [os.path.join(fpath, "{:06d}.jpg".format(center_idx + (i - self.num_frames // 2) * self.stride_frames))
                for i in range(self.num_frames)], center_idx

    def _load_frames(self, paths):
        from PIL import Image
        return self.vid_transform([Image.open(p).convert("RGB") for p in paths])

    def get_frames(self, infos, center_times):
        frames, shifts = [], []
        for n in range(self.num_mix):
            paths, center_idx = self.frame_paths(infos[n], center_times[n])
            shifts.append(center_times[n] - center_idx / float(infos[n][3]))
            frames.append(self._load_frames(paths))
        return frames, shifts

    # ---- music.py:240-252
    def get_ids_labels(self, infos, index, center_times):
        cls = [self.class_int_map[i[5]] for i in infos]
        ids = [os.path.basename(i[0]).split(".")[0][:4] for i in infos]
        name = (str(index) + "_cls" + "_".join(str(c) for c in cls) + "_ids" + "_".join(ids) + "_ct" +
                "_".join(str(round(t)) for t in center_times))
        return name, torch.tensor(cls)

    def __getitem__(self, index):
        random.seed(index)
        infos = self.choose(index)
        audios, mixture, center_times = self.get_audios(infos)
        frames, _ = self.get_frames(infos, center_times)
        name, cls = self.get_ids_labels(infos, index, center_times)
        return {"infos": infos, "audios": [torch.from_numpy(a) for a in audios], "audio_mix": torch.from_numpy(mixture),
                "frames": frames, "id": name, "class": cls}


# ------------------------------------------------------------------------------------------ hand-off
def _stack(ts):
    # No pinning here: with workers > 0 this runs in fork()ed children of a process that may already have
    # initialised HIP, where hipHostMalloc is unusable — and a tensor returned through the worker queue is copied
    # into shared memory anyway.  The parent pins (DataLoader(pin_memory=True): its pin thread runs in the main process).
    return torch.stack(ts, 0)


def collate(samples):
    """default_collate's layout for this dict (lists of per-source tensors stay lists; `infos` transposed like
    default_collate does for nested lists of strings).  GPU-free: safe in forked workers."""
    N = len(samples[0]["audios"])
    return {"audios": [_stack([s["audios"][n] for s in samples]) for n in range(N)],
            "audio_mix": _stack([s["audio_mix"] for s in samples]),
            "frames": [_stack([s["frames"][n] for s in samples]) for n in range(N)],
            "id": [s["id"] for s in samples], "class": torch.stack([s["class"] for s in samples], 0),
            "infos": [[[s["infos"][n][k] for s in samples] for k in range(len(samples[0]["infos"][n]))] for n in range(N)]}


def to_device(batch, device):
    """Asynchronous H2D of the tensors of a collated batch (pinned source -> non_blocking copies on the current stream)."""
    mv = lambda t: t.to(device, non_blocking=True)      # noqa: E731
    out = dict(batch)
    out["audios"] = [mv(t) for t in batch["audios"]]
    out["frames"] = [mv(t) for t in batch["frames"]]
    out["audio_mix"] = mv(batch["audio_mix"])
    return out


def make_loader(csv_paths, args, split, batch_size, shuffle, workers=0, **kw):
    """ConcatDataset of the lists + DataLoader (main.py:633-659); batches are pinned by the parent's pin thread."""
    sets = [MUSICMixDataset(pth, vars(args), split=split, **kw) for pth in csv_paths]
    return torch.utils.data.DataLoader(torch.utils.data.ConcatDataset(sets), batch_size=batch_size, shuffle=shuffle,
                                       num_workers=workers, drop_last=False, collate_fn=collate,
                                       pin_memory=torch.cuda.is_available())
