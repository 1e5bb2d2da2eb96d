"""Host-side label logic of the AD-YOLO path and the synthetic workload generator.

Mirror of the label half of /root/reference/src/datasets.py: grid constants :219-238, ``get_yolo_label``
:457-482 (event -> responsible overlapping grid cells) and ``collate_fn`` :164-184 (rows
``[batch, frame, Gi, Gj, cls, U, V]``).  Pure NumPy/host code -- tiny next to the GPU path -- but
vectorised over events instead of the reference's Python loops.  Synthetic inputs follow SURVEY.md 8d.
"""
import math

import numpy as np
import torch


class YoloLabelEncoder:
    def __init__(self, params=None, grid_size=(45, 45), g_overlap=0.5):
        if params is not None:
            grid_size = params["train_config"]["grid_size"]
            g_overlap = params["train_config"]["g_overlap"]
        gs = np.asarray(grid_size, dtype=np.float64)
        self.nb_grids = [int(math.ceil(360.0 / gs[0])), int(math.ceil(180.0 / gs[1]))]
        half = gs * (0.5 + g_overlap)
        az_c = np.arange(self.nb_grids[0]) * gs[0] - 180.0 + gs[0] * 0.5
        el_c = np.arange(self.nb_grids[1]) * gs[1] - 90.0 + gs[1] * 0.5
        self.az_lb, self.az_ub = az_c - half[0], az_c + half[0]
        self.el_lb = np.clip(el_c - half[1], -90, 90)
        self.el_ub = np.clip(el_c + half[1], -90, 90)

    def encode_events(self, frames, classes, az, el):
        """Vectorised datasets.py:467-480 over E events -> rows (M,6) [frame,Gi,Gj,cls,U,V] (float64).

        Row order matches the reference: event by event, cells in (Gi, Gj) lexicographic order."""
        az = np.asarray(az, dtype=np.float64).copy()
        el = np.asarray(el, dtype=np.float64)
        az[az == 180] = -180.0
        a = az[:, None]
        az_ok = ((self.az_lb[None] <= a) & (a < self.az_ub[None])) | (a + 360 < self.az_ub[None]) | \
                (self.az_lb[None] < a - 360)                               # (E, Gaz)
        e = el[:, None]
        el_ok = (self.el_lb[None] <= e) & (e < self.el_ub[None])            # (E, Gel)
        resp = az_ok[:, :, None] & el_ok[:, None, :]                        # (E, Gaz, Gel)
        ev, gi, gj = np.nonzero(resp)
        frames = np.asarray(frames, dtype=np.float64)
        classes = np.asarray(classes, dtype=np.float64)
        return np.stack([frames[ev], gi.astype(np.float64), gj.astype(np.float64), classes[ev], az[ev], el[ev]], axis=1)

    def get_yolo_label(self, label: dict, nb_label_frames: int):
        """Same signature/result as the reference's ``get_yolo_label`` (list of rows)."""
        fr, cl, az, el = [], [], [], []
        for frame_idx, events in label.items():
            if frame_idx < nb_label_frames:
                for ev in events:
                    fr.append(frame_idx); cl.append(ev[0]); az.append(ev[2]); el.append(ev[3])
        if not fr:
            return []
        rows = self.encode_events(fr, cl, az, el)
        return [[int(r[0]), int(r[1]), int(r[2]), r[3], r[4], r[5]] for r in rows.tolist()]


def collate_fn(batch):
    """datasets.py:164-184: list of (feat, label_rows) -> (feat (B,...), target (M,7) float32).
    Raises, like the reference (torch.cat of an empty list), when no sample has any event."""
    feats, labels = zip(*batch)
    parts = []
    for i, rows in enumerate(labels):
        if len(rows) == 0:
            continue
        r = torch.as_tensor(np.asarray(rows, dtype=np.float32).reshape(len(rows), 6))
        parts.append(torch.cat([torch.full((len(rows), 1), float(i)), r], dim=-1))
    if not parts:
        raise RuntimeError("collate_fn: every sample in the batch has an empty label list")
    return torch.stack([torch.as_tensor(f) for f in feats], 0), torch.cat(parts, 0)


# ------------------------------------------------------------------------------------------- synthetic data
def synthetic_audio(batch, n_samples, seed=1234, device="cpu"):
    """int16-range noise ``round(clip(N(0,0.1)) * 32768)`` -> ``/32768 + 1e-8`` (datasets.py:147),
    layout (B, n_samples, 4) float32 (WAV-native interleaving: one float4 per sample)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = torch.empty(batch, n_samples, 4, dtype=torch.float32)
    for b in range(batch):                      # chunked: keeps host memory bounded at B=64 x 60 s
        x = torch.randn(n_samples, 4, generator=g) * 0.1
        pcm = torch.clamp(torch.round(torch.clamp(x, -1.0, 1.0) * 32768.0), -32768, 32767)
        out[b] = (pcm.double() / 32768.0 + 1e-8).float()
    return out.to(device)


def synthetic_targets(batch, nb_label_frames, nb_classes, seed=1234, encoder=None):
    """Per sample and label frame k ~ {0:.4, 1:.35, 2:.2, 3:.05} events, class ~ U{0..C-1},
    az ~ U[-180,180), el ~ U[-60,60]; expanded by the label encoder (about 4 rows per event).
    -> target (M,7) float32 [b, frame, Gi, Gj, cls, U, V]."""
    rng = np.random.default_rng(seed)
    enc = encoder or YoloLabelEncoder()
    k = rng.choice(4, size=(batch, nb_label_frames), p=[0.4, 0.35, 0.2, 0.05])
    b_idx, f_idx = np.nonzero(k >= 1)
    reps = k[b_idx, f_idx]
    b_ev, f_ev = np.repeat(b_idx, reps), np.repeat(f_idx, reps)
    n = b_ev.shape[0]
    cls = rng.integers(0, nb_classes, size=n)
    az = rng.uniform(-180.0, 180.0, size=n)
    el = rng.uniform(-60.0, 60.0, size=n)
    ev_id = np.arange(n)
    rows = enc.encode_events(ev_id, cls, az, el)            # frame column carries the event id for now
    ids = rows[:, 0].astype(np.int64)
    target = np.concatenate([b_ev[ids, None].astype(np.float64), f_ev[ids, None].astype(np.float64), rows[:, 1:]], axis=1)
    return torch.from_numpy(target.astype(np.float32))


# ------------------------------------------------------------------------------- other --loss label encoders
def _polar_to_xyz(az_deg, el_deg):
    """reference utils/seld_metrics.py:50-65 (convert_output_format_polar_to_cartesian)."""
    el = el_deg * np.pi / 180.0
    az = az_deg * np.pi / 180
    c = np.cos(el)
    return np.cos(az) * c, np.sin(az) * c, np.sin(el)


class ClasswiseLabelEncoder:
    """SEDDOA / ACCDOA / ADPIT frame labels (reference datasets.py:296-348, :350-455).  label: {frame: [[cls, src, az, el]]}."""

    def __init__(self, nb_classes):
        self.nb_classes = nb_classes

    def _sexyz(self, label, nb_label_frames):
        c = self.nb_classes
        se, x, y, z = (np.zeros((nb_label_frames, c)) for _ in range(4))
        for frame, events in label.items():
            if frame < nb_label_frames:
                for ev in events:                           # later events of the same class overwrite (as upstream)
                    ex, ey, ez = _polar_to_xyz(ev[2], ev[3])
                    se[frame, ev[0]], x[frame, ev[0]], y[frame, ev[0]], z[frame, ev[0]] = 1, ex, ey, ez
        return se, x, y, z

    def get_seddoa_label(self, label, nb_label_frames):
        se, x, y, z = self._sexyz(label, nb_label_frames)
        return torch.Tensor(np.concatenate((se, x, y, z), axis=1))

    def get_accdoa_label(self, label, nb_label_frames):
        se, x, y, z = self._sexyz(label, nb_label_frames)
        return torch.Tensor(np.tile(se, 3) * np.concatenate((x, y, z), axis=1))

    def get_adpit_label(self, label, nb_label_frames):
        """(T', 6, 4, C): dummies A0 | B0 B1 | C0 C1 C2 by how many same-class events overlap (1 / 2 / >=3)."""
        c = self.nb_classes
        out = np.zeros((nb_label_frames, 6, 4, c))
        for frame, events in label.items():
            if frame >= nb_label_frames:
                continue
            by_class = {}
            for ev in sorted(events, key=lambda e: e[0]):   # stable sort by class, like list.sort(key=...)
                by_class.setdefault(ev[0], []).append(ev)
            for cls, evs in by_class.items():
                slots = (0,) if len(evs) == 1 else ((1, 2) if len(evs) == 2 else (3, 4, 5))
                for slot, ev in zip(slots, evs):            # at most the first three events of a class are kept
                    ex, ey, ez = _polar_to_xyz(ev[2], ev[3])
                    out[frame, slot, :, cls] = (1.0, ex, ey, ez)
        return torch.Tensor(out)


class AudioStager:
    """WAV int16 -> pinned host buffer -> device -> float32 (reference src/datasets.py:101-107 does ``audio / 32768 + 1e-8``
    on the host in float64 per clip; here the int16 samples cross PCIe (half the bytes of float32) on a side stream and are
    converted by ``adyolo_pcm16_to_f32``).  Double-buffered: ``stage(i)`` of batch k+1 overlaps the step on batch k.

        stager = AudioStager(batch, n_samples, device)
        stager.stage(pcm_list)            # list/array of (n_samples, 4) int16 clips
        audio = stager.get()              # (batch, n_samples, 4) float32 on the device, ordered after the copy
    """

    def __init__(self, batch, n_samples, device="cuda:0", channels=4):
        self.shape = (batch, n_samples, channels)
        self.device = torch.device(device)
        self.host = [torch.empty(self.shape, dtype=torch.int16).pin_memory() for _ in range(2)]
        self.dev = [torch.empty(self.shape, dtype=torch.int16, device=self.device) for _ in range(2)]
        self.stream = torch.cuda.Stream(device=self.device)
        self.events = [torch.cuda.Event(), torch.cuda.Event()]        # copy i has landed
        self.consumed = [None, None]                                  # conversion kernel that read dev[i] (compute stream)
        self.cur = 0

    def stage(self, clips, workers=1):
        """workers > 1: the copy of the clips into the page-locked buffer is split over that many threads (tensor copies release the
        GIL) -- one thread moves 3-5 GB/s of pageable memory, a 64 x 60 s batch is 737 MB: 0.15-0.25 s, longer than the step it feeds."""
        if len(clips) != self.shape[0]:
            raise ValueError("AudioStager: %d clips staged into a buffer of batch %d (a short final batch would leave "
                             "stale audio in the tail rows; use drop_last or a stager of that size)" % (len(clips), self.shape[0]))
        if isinstance(clips, torch.Tensor):
            if tuple(clips.shape) != tuple(self.shape):
                raise ValueError("AudioStager: batch of shape %s, expected %s" % (tuple(clips.shape), tuple(self.shape)))
        else:
            for clip in clips:
                if tuple(clip.shape) != self.shape[1:]:
                    raise ValueError("AudioStager: clip of shape %s, expected %s" % (tuple(clip.shape), self.shape[1:]))
        i = self.cur ^ 1
        h = self.host[i]
        if self.consumed[i] is not None:
            self.events[i].synchronize()          # the previous copy out of this pinned buffer is done
        whole = isinstance(clips, torch.Tensor) and clips.dtype == torch.int16 and not clips.is_cuda

        def fill(lo, hi):
            if whole:                              # one copy call per thread: one GIL round trip instead of one per clip (the
                h[lo:hi].copy_(clips[lo:hi])       # launching thread holds the GIL almost all the time in a launch-bound step)
                return
            for b in range(lo, hi):
                h[b].copy_(torch.as_tensor(np.asarray(clips[b]), dtype=torch.int16))
        nb = len(clips)
        workers = max(1, min(int(workers), nb))
        if workers == 1:
            fill(0, nb)
        else:
            import threading
            step = (nb + workers - 1) // workers
            errors = []                           # an exception inside a worker (dtype / shape / copy error) must not die with
                                                  # its thread: the H2D copy below would ship the PREVIOUS batch's audio

            def guarded(lo, hi):
                try:
                    fill(lo, hi)
                except BaseException as e:        # noqa: BLE001
                    errors.append(e)
            ths = [threading.Thread(target=guarded, args=(lo, min(nb, lo + step))) for lo in range(0, nb, step)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            if errors:
                raise errors[0]
        with torch.cuda.stream(self.stream):
            if self.consumed[i] is not None:
                self.stream.wait_event(self.consumed[i])              # dev[i] has been converted by the compute stream
            self.dev[i].copy_(h, non_blocking=True)
            self.events[i].record(self.stream)
        self.cur = i

    def get(self):
        from . import ops
        cs = torch.cuda.current_stream(self.device)
        cs.wait_event(self.events[self.cur])
        out = ops.pcm16_to_f32(self.dev[self.cur])
        ev = torch.cuda.Event()
        ev.record(cs)
        self.consumed[self.cur] = ev
        return out


class FoaDataset(torch.utils.data.Dataset):
    """Raw-audio counterpart of the reference ``Dataset`` (src/datasets.py:21-162): same constructor, directory layout
    (``foa_dev/dev-train-chunked_<w>s_<s>s``, ``metadata_dev/...``, ``dev-valid`` / ``dev-test``, ``infer_pth``), the
    same per-epoch file sampling without replacement (``sample_filelist_for_train_iter``, ``get_remaining_file`` /
    ``init_remaining_file_from_list`` for resume) and the same CSV label reader -- but ``__getitem__`` stops before the
    arithmetic: it returns ``(pcm int16 (T, 4), comb_no, label_rows)``.  Normalisation, rotation of the audio and the
    features run on the GPU (``AudioStager`` -> ``rotate_audio`` -> ``FeatureExtractor``); the label half of the rotation
    and the AD-YOLO label encoding stay here on the host, as in the reference's DataLoader workers."""

    def __init__(self, params: dict, set_type: str, is_valid=False, rank=None, world=None):
        """rank / world: data-parallel shard (default: torch.distributed if initialised, else RANK / WORLD_SIZE, else 0 / 1).
        Every rank draws the SAME global file list (``batch_size * world * nb_iters`` files from the shared ``random`` seed,
        so ``remaining_file`` stays identical everywhere and the rank-0 checkpoint describes all ranks) and keeps the files
        ``rank, rank + world, ...`` of it: the shards are disjoint and together equal the single-process draw."""
        import copy
        import os
        import random
        self._copy, self._os, self._random = copy, os, random
        if world is None:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized():
                rank, world = tdist.get_rank(), tdist.get_world_size()
            else:
                rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        self.rank, self.world = int(rank or 0), int(world)
        opj = os.path.join
        self.is_valid, self.is_infer, self.set_type = is_valid, set_type == "infer", set_type
        self.loss_nm = params["args"]["loss"]
        dc = params["data_config"]
        # audio format: the reference reads ``foa_dev`` only (datasets.py:36-37,55); ``data_config.audio_format: mic`` selects the
        # DCASE ``mic_dev`` directory of the same layout (4-channel int16 WAVs; BASELINE config 5, features.MicFeatureExtractor)
        adir = {"foa": "foa_dev", "mic": "mic_dev"}[str(dc.get("audio_format", "foa")).lower()]
        if set_type == "train":
            sub = "dev-train-chunked_{}s_{}s".format(dc["chunk_window_s"], dc["chunk_stride_s"])
            self.wav_pth, self.csv_pth = opj(dc["data_pth"], adir, sub), opj(dc["data_pth"], "metadata_dev", sub)
            self.total_filelist = [i.replace(".wav", "") for i in os.listdir(self.wav_pth)]
            self.remaining_file = copy.deepcopy(self.total_filelist)
            self.nb_samples = params["train_config"]["batch_size"] * params["train_config"]["nb_iters"] * self.world
            self.filelist = []
            self.sample_filelist_for_train_iter()
        else:
            if self.is_infer:
                self.wav_pth, self.csv_pth = str(params["args"]["infer_pth"]), None
            else:
                self.wav_pth = opj(dc["data_pth"], adir, "dev-{}".format(set_type))
                self.csv_pth = opj(dc["data_pth"], "metadata_dev", "dev-{}".format(set_type))
            self.filelist = [i.replace(".wav", "") for i in os.listdir(self.wav_pth)]
            if self.world > 1:                   # evaluation files are dealt round-robin over the ranks (sorted: listdir order is not a contract)
                self.filelist = sorted(self.filelist)[self.rank::self.world]
        self.hop_label = int(dc.get("sr", 24000) * dc.get("label_hop_len_s", 0.1))
        self.rotate = bool(params.get("aug_config", {}).get("rotation_augment", False)) and not is_valid
        if self.loss_nm != "adyolo":
            raise NotImplementedError("FoaDataset encodes AD-YOLO labels; use ClasswiseLabelEncoder for %s" % self.loss_nm)
        self.encoder = YoloLabelEncoder(params)

    def sample_filelist_for_train_iter(self):
        """datasets.py:67-91, statement for statement (so that a seeded ``random`` draws the same files)."""
        copy, random = self._copy, self._random
        self.filelist = []
        if len(self.remaining_file) >= self.nb_samples:
            self.filelist = random.sample(self.remaining_file, self.nb_samples)
            for fnm in self.filelist:
                self.remaining_file.remove(fnm)
        elif len(self.remaining_file) <= 0:
            self.remaining_file = copy.deepcopy(self.total_filelist)
            self.filelist = random.sample(self.remaining_file, self.nb_samples)
            for fnm in self.filelist:
                self.remaining_file.remove(fnm)
        else:
            random.shuffle(self.remaining_file)
            pre_sampled = copy.deepcopy(self.remaining_file)
            self.remaining_file = copy.deepcopy(self.total_filelist)
            self.filelist = random.sample(self.remaining_file, self.nb_samples - len(pre_sampled))
            for fnm in self.filelist:
                self.remaining_file.remove(fnm)
            self.filelist.extend(pre_sampled)
        if self.world > 1:                       # this rank's shard of the global draw
            self.filelist = self.filelist[self.rank::self.world]

    def init_remaining_file_from_list(self, remaining_file: list):
        self.remaining_file = remaining_file

    def get_remaining_file(self):
        return self.remaining_file

    def get_filelist(self):
        return self.filelist

    @staticmethod
    def load_csv2dict(csv_pth):
        """datasets.py:103-117: ``frame,class,source,azimuth,elevation`` (or cartesian x,y,z) rows -> {frame: [[...]]}."""
        label = {}
        with open(csv_pth, "r") as fid:
            for line in fid:
                words = line.strip().split(",")
                if len(words) < 5:
                    continue
                frame_idx = int(words[0])
                label.setdefault(frame_idx, []).append([int(words[1]), int(words[2])] + [float(w) for w in words[3:6]])
        return label

    def __len__(self):
        return len(self.filelist)

    def __getitem__(self, index):
        from scipy.io import wavfile
        from .augmentations import rotate_labels
        name = self.filelist[index]
        _, pcm = wavfile.read(self._os.path.join(self.wav_pth, name + ".wav"))          # int16 (T, 4)
        label = {} if self.is_infer else self.load_csv2dict(self._os.path.join(self.csv_pth, name + ".csv"))
        comb_no = 0
        if self.rotate:
            comb_no = int(self._random.uniform(0, 16))                                    # augmentations.py:76
            label = rotate_labels(label, comb_no)
        nb_label_frames = pcm.shape[0] // self.hop_label
        return np.ascontiguousarray(pcm, dtype=np.int16), comb_no, self.encoder.get_yolo_label(label, nb_label_frames)


def audio_collate_fn(batch):
    """list of FoaDataset items -> (pcm int16 (B, T, 4) host tensor, comb_nos list, target (M, 7) float32); the target
    rows are built exactly like ``collate_fn`` (datasets.py:164-184)."""
    pcms, combs, labels = zip(*batch)
    _, target = collate_fn([(np.zeros(1, dtype=np.float32), rows) for rows in labels])
    return torch.stack([torch.from_numpy(p) for p in pcms], 0), list(combs), target
