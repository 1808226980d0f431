"""Drop-in for the reference's utils/nlos_pose_dataloader.py:20-148 (`NlosPoseDataset`) with the per-sample
arithmetic on the GPU: the CPU worker only reads files (and expands the .hdr run-length container with the
library's host routine); decode, both normalisations, gray conversion, crop and the averaging pyramid run in
csrc/ingest_kernels.hip through the C ABI (hp_rgbe_decode, hp_ingest_rgbe_to_meas, hp_box_downsample_round).

`__getitem__` returns what the reference returns -- (meas (1,T,H,W), vol (1,D,H,W), joints (24,3), person_id) --
except that meas and vol are fp32 tensors on `device` instead of NumPy arrays.  Error behaviour follows
:72-104: a sample whose file cannot be read or whose maximum is below 1e-10 is replaced by sample 0 (the
reference also swaps the joints but keeps the failing sample's volume; so does this class).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib

FRAMES, KEEP_FRAMES = 600, 512  # rearrange '(t h) w -> t h w', t=600 and [:512]  (:107)


def decode_hdr(path: str, pinned=None):
    """Radiance file -> (rows, W, 4) uint8 RGBE on the host (file parsing only; hp_rgbe_decode).  `pinned`: a callable
    nbytes -> page-locked uint8 torch tensor to decode into (the result is then a view of it, a torch tensor).
    The first call is the size query: it parses the text header only and returns before the run-length stream."""
    L = _lib.lib()
    raw = np.fromfile(path, dtype=np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    _lib.check(L.hp_rgbe_decode(raw.ctypes.data, raw.size, C.byref(w), C.byref(h), None, 0), "hp_rgbe_decode")
    n = h.value * w.value * 4
    if pinned is not None:
        buf = pinned(n)
        _lib.check(L.hp_rgbe_decode(raw.ctypes.data, raw.size, C.byref(w), C.byref(h), buf.data_ptr(), n), "hp_rgbe_decode")
        return buf[:n].view(h.value, w.value, 4)
    out = np.empty((h.value, w.value, 4), np.uint8)
    _lib.check(L.hp_rgbe_decode(raw.ctypes.data, raw.size, C.byref(w), C.byref(h), out.ctypes.data, out.nbytes), "hp_rgbe_decode")
    return out


def rgbe_to_meas(rgbe: torch.Tensor, downsample_cnt: int, frames: int = FRAMES, keep: int = KEEP_FRAMES) -> torch.Tensor:
    """(frames*H, W, 4) uint8 device tensor -> (keep/2^(cnt+1), H/2^cnt, W/2^cnt) fp32 (:74-83, :107-117)."""
    if not rgbe.is_cuda:
        raise _lib.HiddenPoseHipError("rgbe_to_meas needs a HIP device tensor (no CPU path)")
    rows, W, four = rgbe.shape
    assert four == 4 and rgbe.dtype == torch.uint8 and rows % frames == 0
    H = rows // frames
    div = 1 << downsample_cnt
    meas = torch.empty(keep // (2 * div), H // div, W // div, dtype=torch.float32, device=rgbe.device)
    maxima = torch.empty(2, dtype=torch.float32, device=rgbe.device)
    rgbe = rgbe.contiguous()
    st = torch.cuda.current_stream(rgbe.device).cuda_stream
    _lib.check(_lib.lib().hp_ingest_rgbe_to_meas(rgbe.data_ptr(), frames, H, W, keep, downsample_cnt, meas.data_ptr(),
                                                 maxima.data_ptr(), st), "hp_ingest_rgbe_to_meas")
    if abs(float(maxima[0])) < 1e-10:  # :75 (one 4-byte read-back per sample)
        raise ValueError("wrong Meas File!")
    return meas


def box_pyramid(v: torch.Tensor, rounds: int) -> torch.Tensor:
    """:114-121: `rounds` times pair averages along each of the three axes of a (D,H,W) fp32 device tensor."""
    if not v.is_cuda:
        raise _lib.HiddenPoseHipError("box_pyramid needs a HIP device tensor (no CPU path)")
    L = _lib.lib()
    st = torch.cuda.current_stream(v.device).cuda_stream
    for _ in range(rounds):
        D, H, W = v.shape
        out = torch.empty(D // 2, H // 2, W // 2, dtype=torch.float32, device=v.device)
        _lib.check(L.hp_box_downsample_round(v.data_ptr(), out.data_ptr(), D, H, W, *v.stride(), st), "hp_box_downsample_round")
        v = out
    return v


def remap_joints(joints: np.ndarray, vol_size: int, heatmap_size: int) -> np.ndarray:
    """:128-143: metres -> 256^3 voxels, (x,y,z) -> (d,h,w), then to heat-map voxels.  24x3 values: host."""
    j = np.asarray(joints, dtype=np.float64)
    w = j[:, 0] * 128 + 128
    h = 256 - (j[:, 1] * 128 + 128)
    d = 225 - (j[:, 2] * 128 + 128)
    return np.stack([d, h, w], axis=1) / (vol_size / heatmap_size)


class NlosPoseDataset(Dataset):
    def __init__(self, cfg, datapath, device=None):
        super().__init__()
        self.vol_size = cfg.DATASET.VOL_SIZE
        self.heatmap_size = cfg.MODEL.HEATMAP_SIZE
        self.downsample_cnt = cfg.DATASET.DAWNSAMPLE_CNT
        self.phase = cfg.DATASET.PHASE
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.measFiles, self.volFiles, self.jointsFiles, self.wrongMeasFiles = [], [], [], []
        for pose in os.listdir(datapath):
            for split in os.listdir(os.path.join(datapath, pose)):
                if self.phase not in split:
                    continue
                base = os.path.join(datapath, pose, split)
                for name in os.listdir(os.path.join(base, "meas")):
                    stem, ext = os.path.splitext(name)
                    assert ext == ".hdr", f"Data type should be .hdr,not {name} in {os.path.join(base, 'meas')}"
                    vol = os.path.join(base, "vol", stem + ".mat")
                    joints = os.path.join(base, "joints", stem + ".joints")
                    assert os.path.isfile(vol), f"Do not have related vol {vol}"
                    assert os.path.isfile(joints), f"Do not have related joints {joints}"
                    self.measFiles.append(os.path.join(base, "meas", name))
                    self.volFiles.append(vol)
                    self.jointsFiles.append(joints)
        for kind, files in (("meas", self.measFiles), ("vol", self.volFiles), ("joints", self.jointsFiles)):
            print(f"total {self.phase} {kind} is {len(files)}")

    # ---- host half (file I/O + run-length expansion; runs in worker threads of PrefetchingLoader) ----------------
    def wrong_meas(self, index):
        """:76-81, :96-104: report a measurement that cannot be used; the caller substitutes sample 0."""
        meas_file = self.measFiles[index]
        print(f"--------------------\nNo.{index} {meas_file} meas is wrong. \n--------------------------\n")
        self.wrongMeasFiles.append(meas_file)

    def load_host(self, index, pinned=None):
        """Everything of sample `index` that needs no GPU: the expanded RGBE image (uint8, pinned if `pinned` hands a
        reusable page-locked buffer), the raw volume, the remapped joints.  As in the reference (:72-104) only the
        MEASUREMENT is guarded: a file that cannot be decoded is replaced by sample 0's measurement and joints while the
        volume stays the failing sample's; an unreadable .mat / .joints file raises (loadmat / loadtxt at :108-109 are
        outside the reference's try block)."""
        from scipy.io import loadmat

        meas_index = index
        try:
            rgbe = decode_hdr(self.measFiles[index], pinned)
        except Exception:
            self.wrong_meas(index)
            meas_index = 0
            rgbe = decode_hdr(self.measFiles[0], pinned)
        meas_file, joint_file = self.measFiles[meas_index], self.jointsFiles[meas_index]
        vol = loadmat(self.volFiles[index])["vol"].astype(np.float32)
        joints = remap_joints(np.loadtxt(joint_file), self.vol_size[0], self.heatmap_size[0])
        return {"rgbe": rgbe, "vol": vol, "joints": joints, "id": os.path.splitext(os.path.basename(meas_file))[0],
                "index": index}

    # ---- device half (decode, normalisations, gray, crop, pyramids; on the CURRENT stream) ---------------------------
    def meas_to_device(self, rgbe: torch.Tensor, index: int = 0) -> torch.Tensor:
        """Expanded RGBE image (device, uint8) -> network measurement (:74-83, :107-117).  Raises ValueError on an
        all-zero image (:75).  Subclasses (the noise dataset) replace this stage."""
        return rgbe_to_meas(rgbe, self.downsample_cnt)

    def to_device(self, host):
        rgbe = host["rgbe"] if torch.is_tensor(host["rgbe"]) else torch.from_numpy(host["rgbe"])
        meas = self.meas_to_device(rgbe.to(self.device, non_blocking=True), host.get("index", 0))
        vol = box_pyramid(torch.from_numpy(host["vol"]).to(self.device, non_blocking=True), self.downsample_cnt)
        return meas[None], vol[None]

    def __getitem__(self, index):
        host = self.load_host(index)
        try:
            meas, vol = self.to_device(host)
        except ValueError:   # :75-81: an all-zero measurement -> sample 0's measurement and joints, this sample's volume
            self.wrong_meas(index)
            h0 = self.load_host(0)
            host = dict(h0, vol=host["vol"], index=index)
            meas, vol = self.to_device(host)
        return meas, vol, host["joints"], host["id"]

    def __len__(self):
        return len(self.volFiles)


class PrefetchingLoader:
    """Batches of a NlosPoseDataset with the ingest overlapped with training (reference: DataLoader(num_workers=8,
    pin_memory=True), train.py:119-122, whose workers decode on the CPU).

    * `workers` host threads read the files and expand the Radiance run-length container (hp_rgbe_decode releases the
      GIL) straight into recycled page-locked buffers;
    * one device thread copies each expanded image to the GPU and runs the ingest kernels on a SIDE stream, stacks the
      batch and records an event;
    * the consumer's stream waits on that event only -- at 128^3 the 157 MB image of a sample otherwise serialises
      file read + expansion + H2D with the training step (~26 samples/s).
    `depth` batches are kept ready.  Sample order = `sampler` if given, else range(len) (shuffled with `seed` + epoch
    when `shuffle`).  A measurement that cannot be read (or is all zero) is replaced by sample 0 as in the reference."""

    def __init__(self, dataset, batch_size, sampler=None, shuffle=False, drop_last=True, depth=2, workers=4, seed=410):
        self.ds, self.bs, self.sampler, self.shuffle, self.drop_last = dataset, int(batch_size), sampler, shuffle, drop_last
        self.depth, self.workers, self.seed, self.epoch = max(1, depth), max(1, workers), seed, 0
        self._free = []          # recycled pinned buffers
        import threading

        self._lock = threading.Lock()

    def set_epoch(self, epoch):
        self.epoch = int(epoch)
        if self.sampler is not None and hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(epoch)

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def _indices(self):
        if self.sampler is not None:
            return list(iter(self.sampler))
        n = len(self.ds)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            return torch.randperm(n, generator=g).tolist()
        return list(range(n))

    def _pinned(self, nbytes):
        with self._lock:
            for i, t in enumerate(self._free):
                if t.numel() >= nbytes:
                    return self._free.pop(i)
        return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)

    def _recycle(self, t):
        base = t._base if t._base is not None else t
        with self._lock:
            if len(self._free) < self.depth * self.bs + self.workers:
                self._free.append(base.reshape(-1))

    def __iter__(self):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor

        ds, dev = self.ds, self.ds.device
        idx = self._indices()
        nb = len(self)
        batches = [idx[i * self.bs:(i + 1) * self.bs] for i in range(nb)]
        ready = queue.Queue(maxsize=self.depth)
        pool = ThreadPoolExecutor(max_workers=self.workers)
        stop = threading.Event()

        def host(i):
            return ds.load_host(i, self._pinned)   # guards the measurement decode only; .mat / .joints errors surface

        def device_stage():
            side = torch.cuda.Stream(dev)
            try:
                # host loads are submitted `depth` batches ahead of the device stage
                futs = {}
                nxt = 0
                ahead = self.depth + 1
                for bi, batch in enumerate(batches):
                    while nxt < min(len(batches), bi + ahead):
                        futs[nxt] = [pool.submit(host, i) for i in batches[nxt]]
                        nxt += 1
                    items = [f.result() for f in futs.pop(bi)]
                    if stop.is_set():
                        return
                    with torch.cuda.device(dev), torch.cuda.stream(side):
                        ms, vs = [], []
                        for k, h in enumerate(items):
                            try:
                                m, v = ds.to_device(h)
                            except ValueError:   # all-zero measurement (:75): sample 0's measurement and joints instead
                                ds.wrong_meas(h["index"])
                                h0 = host(0)
                                items[k] = dict(h0, vol=h["vol"], index=h["index"])
                                m, v = ds.to_device(items[k])
                            ms.append(m)
                            vs.append(v)
                        meas, vol = torch.stack(ms), torch.stack(vs)
                        joints = torch.stack([torch.as_tensor(h["joints"], dtype=torch.float32) for h in items]).to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    ids = [h["id"] for h in items]
                    side.synchronize()     # the pinned buffers may be reused once their copies are done
                    for h in items:
                        if torch.is_tensor(h["rgbe"]):
                            self._recycle(h["rgbe"])
                    ready.put((meas, vol, joints, ids, ev))
                ready.put(None)
            except BaseException as e:   # surface worker failures in the consumer
                ready.put(e)

        th = threading.Thread(target=device_stage, daemon=True)
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                meas, vol, joints, ids, ev = item
                torch.cuda.current_stream(dev).wait_event(ev)
                for t in (meas, vol, joints):
                    t.record_stream(torch.cuda.current_stream(dev))
                yield meas, vol, joints, ids
        finally:
            stop.set()
            while th.is_alive():   # drain so that the producer can finish
                try:
                    ready.get(timeout=0.1)
                except queue.Empty:
                    pass
            pool.shutdown(wait=False)
