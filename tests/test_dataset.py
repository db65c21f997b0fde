"""Input-pipeline contract: the loader's sampling rules against the fixture generated from the
reference's own MUSICMixDataset (tests/golden/dataset.json, made by oracle/gen_golden.py `dataset`), and an
end-to-end run on a synthetic on-disk dataset (wav + jpg files written here).  CPU only."""
import json
import os
import random
import zlib

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _raw(path, center_t, n):      # the decoder stand-in the fixture was generated with (oracle/gen_golden.py)
    rs = np.random.RandomState(zlib.crc32(("%s|%.6f" % (path, center_t)).encode()) & 0x7fffffff)
    return (rs.rand(n).astype(np.float32) - 0.5) * 2.4


def test_sampling_rules_match_the_reference_fixture():
    import avsep_amd as P
    from avsep_amd import dataset as PD

    class DS(PD.MUSICMixDataset):
        def _load_audio_file(self, path, center_t):
            return _raw(path, center_t, int((self.margin * 2 + self.audSec) * self.audRate)), self.audRate

        def _load_frames(self, paths):
            self.seen.append(list(paths))
            return torch.zeros(1)
    with open(os.path.join(GOLD, "dataset.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 5
    for c in cases:
        a = P.ArgParser().parse_train_arguments(c["argv"], verbose=False)
        ds = DS(os.path.join(GOLD, "dataset_list.csv"), vars(a), split=c["split"], **c["kw"])
        assert len(ds) == c["len"] and ds.list_samples[:3] == c["first_rows"]      # repeat + seeded shuffle
        for rec in c["items"]:
            ds.seen = []
            it = ds[rec["index"]]
            assert [list(i) for i in it["infos"]] == rec["infos"]
            assert it["id"] == rec["id"] and it["class"].tolist() == rec["class"]
            assert ds.seen == rec["frame_paths"]
            for n, aud in enumerate(it["audios"]):
                assert aud.dtype == torch.float32 and aud.shape == (a.audLen,)
                assert abs(float(aud.double().sum()) - rec["audio_sum"][n]) < 1e-6
                assert abs(float(aud.double().abs().sum()) - rec["audio_abs"][n]) < 1e-6
            assert abs(float(it["audio_mix"].double().abs().sum()) - rec["mix_abs"]) < 1e-6


def _make_disk_dataset(root, rate=11025, secs=9.0, fps=4.0, size=(160, 120)):
    from PIL import Image
    from scipy.io import wavfile
    rows, rs = [], np.random.RandomState(3)
    from avsep_amd.dataset import MUSIC11_CLASSES
    for ci, cls in enumerate(MUSIC11_CLASSES):   # the "dc" rule draws among all 11 classes: each needs a clip
        for k in range(1):
            vid = f"{cls[:3]}{k:02d}xyz"
            apath, fdir = os.path.join(root, "audio", cls, vid + ".wav"), os.path.join(root, "frames", cls, vid + ".mp4")
            os.makedirs(os.path.dirname(apath), exist_ok=True)
            os.makedirs(fdir, exist_ok=True)
            t = np.arange(int(secs * rate)) / rate
            wav = 0.4 * np.sin(2 * np.pi * (110.0 * (ci + 1) + 30 * k) * t)
            wavfile.write(apath, rate, (wav * 32767).astype(np.int16))
            nf = int(secs * fps)
            for i in range(nf + 1):
                Image.fromarray(rs.randint(0, 255, (size[1], size[0], 3), dtype=np.uint8)).save(
                    os.path.join(fdir, "{:06d}.jpg".format(i)), quality=60)
            rows.append([apath, fdir, str(nf), str(fps), str(secs), cls])
    lst = os.path.join(root, "list.csv")
    with open(lst, "w") as f:
        f.write("\n".join(",".join(r) for r in rows) + "\n")
    return lst


@pytest.mark.parametrize("split", ["train", "val"])
def test_loader_end_to_end_on_disk(tmp_path, split):
    import avsep_amd as P
    from avsep_amd import dataset as PD
    lst = _make_disk_dataset(str(tmp_path))
    a = P.ArgParser().parse_train_arguments(
        ["--num_frames", "3", "--stride_frames", "2", "--imgSize", "64", "--audLen", "16383", "--margin", "1.0",
         "--train_repeat", "2", "--val_repeat", "2"], verbose=False)
    loader = PD.make_loader([lst], a, split, batch_size=4, shuffle=False)
    assert len(loader.dataset) == 22
    batch = next(iter(loader))
    assert [t.shape for t in batch["audios"]] == [(4, 16383)] * 2 and batch["audio_mix"].shape == (4, 16383)
    assert [t.shape for t in batch["frames"]] == [(4, 3, 3, 64, 64)] * 2 and batch["class"].shape == (4, 2)
    assert len(batch["id"]) == 4 and len(batch["infos"]) == 2 and len(batch["infos"][0]) == 6
    # the mixture is the sum of the (already /N) sources; sources are real audio (not silence), within +-1
    assert torch.allclose(batch["audio_mix"], batch["audios"][0] + batch["audios"][1], atol=1e-7)
    assert batch["audios"][0].abs().max() <= 0.5 + 1e-6 and batch["audios"][0].abs().max() > 0.05
    # partners come from a different class (rate_dc = 1)
    assert all(batch["class"][b, 0] != batch["class"][b, 1] for b in range(4))
    # per-index determinism (random.seed(index)), whatever was drawn before
    ds = loader.dataset.datasets[0]
    random.seed(999)
    again = ds[1]
    assert torch.equal(again["audio_mix"], batch["audio_mix"][1]) and torch.equal(again["frames"][1], batch["frames"][1][1])
    # ImageNet normalisation: channel 0 of a 0..255 image lies within (0-0.485)/0.229 .. (1-0.485)/0.229
    f = batch["frames"][0]
    assert f[:, 0].min() >= -0.485 / 0.229 - 1e-5 and f[:, 0].max() <= (1 - 0.485) / 0.229 + 1e-5
    if split == "val":   # centre crop of the bicubic resize to 64 on the short side: recompute one frame by hand
        from PIL import Image
        info, name = again["infos"][0], again["id"]
        t0 = None
        random.seed(1)
        infos = ds.choose(1)
        _, _, cts = ds.get_audios(infos)
        paths, _ = ds.frame_paths(infos[0], cts[0])
        img = Image.open(paths[1]).convert("RGB").resize((85, 64), Image.BICUBIC)       # 160x120 -> 85x64
        crop = img.crop((10, 0, 74, 64))                                                # int(round((85-64)/2)) = 10
        ref = (torch.from_numpy(np.asarray(crop).copy()).permute(2, 0, 1).float() / 255.0 -
               torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
        assert torch.allclose(again["frames"][0][:, 1], ref, atol=1e-6)


def test_wav_segment_reader(tmp_path):
    from scipy.io import wavfile
    from avsep_amd.dataset import read_wav_segment
    rate = 22050
    t = np.arange(rate * 3) / rate
    stereo = np.stack([np.sin(2 * np.pi * 440 * t), np.zeros_like(t)], 1)
    p = str(tmp_path / "a.wav")
    wavfile.write(p, rate, (stereo * 32767).astype(np.int16))
    seg, sr = read_wav_segment(p, 11025, 1.0, 1.0)                 # stereo -> mono, 22050 -> 11025
    assert sr == 11025 and abs(len(seg) - 11025) <= 1 and seg.dtype == np.float32
    ref = 0.5 * np.sin(2 * np.pi * 440 * (1.0 + np.arange(len(seg)) / 11025.0))
    assert np.abs(seg[200:-200] - ref[200:-200]).max() < 2e-3
    seg2, _ = read_wav_segment(p, 22050, 0.5, 0.25)
    assert len(seg2) == int(round(0.75 * rate)) - int(round(0.5 * rate))
