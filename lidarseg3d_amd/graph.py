"""One inference frame = one hipGraph ("avoid host syncs for V"; the reference's own per-frame host round trips are
det3d/ops/voxel/src/scatter_points_cuda.cu:218-221 and the dataloader's CPU voxelization).

A frame of this path is ~220 small and large kernel launches on two HIP streams.  Submitted from Python they cost ~10 us of host time
each, and the latency-bound geometry chain (4 strided rulebooks, 4 SubM rulebooks, 4 tile plans: ~100 launches of 2-80 us) is then
bound by the host, not by the GPU: the convolutions of level 2 wait ~0.6 ms per frame for it (profiles/round3_timeline_eager.txt).
Capacity mode (detectors.CAPACITY_MODE) removed every host synchronisation from the frame and gave every tensor a shape that does not
depend on the data; that makes the whole frame capturable: FrameGraph records one capacity-mode forward - both streams - into a
hipGraph (torch.cuda.CUDAGraph: PyTorch is the plumbing for streams and memory here) and replays it per frame.

    fg = FrameGraph(model, example)      # example: the collated dict with `points` [N, 1 + C] on the device (+ the camera inputs of MSeg3D)
    ret = fg(example)                    # same list of dicts as model(example, return_loss=False)

Correctness contract: a replay runs exactly the launches of the eager capacity-mode frame (bit-identical results).  The capacities of the
strided rulebooks are the ones learned from the warm-up frames (1.3x the largest count seen); every replay copies the frame's counts and
overflow flags to pinned host memory, and __call__ checks them after the frame: an overflowing frame is computed again by the eager
path (host-side counts, always correct) and the graph is captured again with the new capacities.  Inputs of another shape than the
captured one go to the eager path as well (a graph is a fixed-shape object) - unless the graph was captured with `point_keys`: then a
frame of FEWER points than the captured capacity is padded inside the graph's input buffers with rows that belong to no frame (batch
index = batch_size, coordinates far outside every range: the voxelizer rejects them, the devoxelization and the decoder never visit them,
the per-point tail gives them zero features) and the outputs are cut back to the frame's rows - bit-identical labels for the real points.
BucketedFrameGraph keeps one such graph per point-count bucket, so a sweep stream whose point count changes from frame to frame
(tools/dist_test.py:189-230 times exactly that) stays on the graph path.  A batch of several
frames (round 4) is captured up to the labels of all points; the split into per-frame results - boolean masks, i.e. host synchronisations - runs
after the replay.

Several frames in flight: one FrameGraph per frame slot, each captured on its OWN stream (`stream=`: the arrival counters of the tile
kernel's channel split are per stream, the pinned count buffer is per capture), then per slot `with torch.cuda.stream(s): fg.launch(ex)`
and later `fg.finish(ex)` - the replays of different slots overlap on the GPU (bench.py's `throughput_mode.graph_*`)."""
import torch

from . import detectors, ops


PAD_COORD = 1.0e6  # metres: outside any point-cloud range, finite in f32 arithmetic

# Captured graphs are RETIRED, never destroyed.  A frame's graph forks into side streams (geometry, lateral blocks, head), so its hipGraphExec runs on
# several internal streams of the HIP runtime; on ROCm 7.0 destroying a hipGraphExec while another one exists leaves the runtime with dangling
# stream pointers, and the replay of a graph captured LATER segfaults in hip::Graph::UpdateStreams (hipGraphLaunch) - reproduced in seconds by
# tools/scratch/stress_bucket.py (two bucket graphs of one model captured, replayed, dropped; the next model's first replay crashes), gone when the
# torch.cuda.CUDAGraph objects stay referenced, whatever else is freed.  The price is the retired graph's private memory pool (1 - 3 GB for a 120k-point
# frame) until the process ends; graphs retire only on an overflow recapture, an LRU eviction of BucketedFrameGraph or when their FrameGraph is dropped.
_RETIRED = []


def retired_graphs():
    """number of captured graphs that were dropped by their owners and are kept alive (see _RETIRED)"""
    return len(_RETIRED)


def pad_rows(key, like, rows, batch_size):
    """`rows` padding rows for the per-point input `key`: points = (batch_size, far away, zero features); any other per-point table = zeros
    (points_cuv: valid flag 0 - no camera sees the point)"""
    pad = torch.zeros((rows,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
    if key == "points" and rows:
        pad[:, 0] = float(batch_size)
        pad[:, 1:4] = PAD_COORD
    return pad


class FrameGraph(object):
    def __init__(self, model, example, warmup=3, stream=None, point_keys=None, pool=None):
        if model.training:
            raise ValueError("FrameGraph is an inference path: model.eval() first")
        if (getattr(model, "test_cfg", None) or {}).get("tta_flag", False) and int(example.get("batch_size", 1)) != 1:
            raise ValueError("FrameGraph does not capture test-time-augmentation batches (their merge is per group of frames)")
        if not detectors.CAPACITY_MODE:
            raise ValueError("FrameGraph needs capacity mode (LS3D_CAPACITY_MODE=0 is set)")
        self.model, self.warmup, self.stream, self.pool = model, int(warmup), stream, pool
        self.batch_size = int(example.get("batch_size", 1))
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example.items()}
        if self.batch_size > 1:
            # predict() splits a batch into frames with boolean masks (a host synchronisation each): the captured forward ends at the labels
            # of all points, _after_replay() splits them by the batch column of the replayed example
            self.static["_unsplit_predict"] = True
        self.shapes = {k: (tuple(v.shape), v.dtype) for k, v in example.items() if torch.is_tensor(v)}
        # per-point inputs (dim 0 = the sweep's points): a frame may bring FEWER rows than captured, the rest is padding (module docstring)
        self.point_keys = tuple(k for k in (point_keys or ()) if k in self.shapes)
        self.capacity = self.shapes[self.point_keys[0]][0][0] if self.point_keys else None
        self.graph, self.ret, self.record, self.recaptures, self.fallbacks = None, None, None, 0, 0
        self._rows = self.capacity
        self._capture()

    def _capture(self):
        model, bb = self.model, self.model.backbone
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # packs the weights, sets the kernels' attributes, learns the capacities
                model(dict(self.static), return_loss=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bb.__dict__.pop("_captured_record", None)
        old = bb.__dict__.get("_pinned_counts")
        if old is not None:  # this graph's own pinned count buffer (allocated outside the capture): graphs of other frame slots keep theirs
            bb.__dict__["_pinned_counts"] = ops.registered_host(old.shape, old.dtype)
        if self.graph is not None:
            _RETIRED.append(self.graph)  # the graph this capture replaces (overflow -> recapture)
        g = torch.cuda.CUDAGraph()
        kw = dict(pool=self.pool) if self.pool is not None else {}
        with torch.no_grad(), (torch.cuda.graph(g, **kw) if self.stream is None else torch.cuda.graph(g, stream=self.stream, **kw)):
            ret = model(dict(self.static), return_loss=False)
        rec = bb.__dict__.pop("_captured_record", None)
        if rec is None:
            raise RuntimeError("the captured forward did not take the capacity path (see detectors._capacity_ok)")
        self.graph, self.ret, self.record = g, ret, rec
        self.logits = model.point_head.forward_ret_dict.get("out_logits")

    def __del__(self):
        g = self.__dict__.get("graph")
        if g is not None and _RETIRED is not None:  # (None: the interpreter is shutting down and has cleared the module's globals)
            _RETIRED.append(g)

    def matches(self, example):
        if int(example.get("batch_size", 1)) != self.batch_size:
            return False
        n = example[self.point_keys[0]].shape[0] if (self.point_keys and self.point_keys[0] in example) else None
        for k, (s, d) in self.shapes.items():
            v = example.get(k)
            if not torch.is_tensor(v) or v.dtype != d:
                return False
            if k in self.point_keys:
                if tuple(v.shape[1:]) != s[1:] or v.shape[0] != n or n > s[0]:
                    return False
            elif tuple(v.shape) != s:
                return False
        return True

    _shared_replay_stream = {}  # device -> handle of the stream the graphs captured WITHOUT `stream=` are replayed on

    def launch(self, example):
        """copy the inputs and replay on the CURRENT stream, without waiting (the example must match the captured shapes); finish() next"""
        cur = torch.cuda.current_stream()
        if self.stream is None:
            # graphs captured on torch's shared capture stream have the SAME arrival counters of the tile kernel's channel split baked in
            # (ops keys them by the capture stream): replayed side by side on two streams they would race on the counters' parity
            seen = FrameGraph._shared_replay_stream.setdefault(cur.device, cur.cuda_stream)
            if seen != cur.cuda_stream:
                raise RuntimeError("FrameGraphs captured without stream= share the tile kernel's arrival counters and must be replayed on ONE "
                                   "stream; capture each frame slot with its own stream= to run them side by side")
        elif cur.cuda_stream != self.stream.cuda_stream:
            raise RuntimeError("a FrameGraph captured with stream= is replayed on that stream (its arrival counters belong to it)")
        self._rows = example[self.point_keys[0]].shape[0] if self.point_keys else None
        for k in self.shapes:
            if k in self.point_keys and self._rows < self.capacity:
                self.static[k][:self._rows].copy_(example[k], non_blocking=True)
                self.static[k][self._rows:] = self._pad(k)[:self.capacity - self._rows]
            else:
                self.static[k].copy_(example[k], non_blocking=True)
        self.graph.replay()
        self._launched_on = torch.cuda.current_stream()

    def _pad(self, key):
        cache = self.__dict__.setdefault("_pad_cache", {})
        if key not in cache:
            cache[key] = pad_rows(key, self.static[key], self.capacity, self.batch_size)
        return cache[key]

    def finish(self, example, clone=True):
        """wait for the frame launch() started, check its rulebook counts (overflow -> eager rerun + recapture) -> the frame's outputs"""
        self._launched_on.synchronize()  # the frame is done: its counts are on the host, its outputs can be handed out
        return self._after_replay(example, clone)

    def __call__(self, example, clone=True):
        """-> model(example, return_loss=False).  clone=False returns the graph's own output tensors (overwritten by the next call)"""
        if not self.matches(example):
            self.fallbacks += 1
            with torch.no_grad():
                return self.model(example, return_loss=False)
        self.launch(example)
        return self.finish(example, clone)

    def _after_replay(self, example, clone):
        host, key = self.record
        if not self.model.backbone.apply_counts(host.tolist(), key):
            # a rulebook overflowed the captured capacity: this frame on host-side counts, then a new graph on relearned capacities
            self.fallbacks += 1
            cap, detectors.CAPACITY_MODE = detectors.CAPACITY_MODE, False
            try:
                with torch.no_grad():
                    out = self.model(example, return_loss=False)
            finally:
                detectors.CAPACITY_MODE = cap
            self.recaptures += 1
            self._capture()
            return out
        # the captured dicts hold the FIRST frame's non-tensor fields: every replay hands out the current frame's `metadata` (the reference keys
        # its saved predictions by output['metadata']['token'], tools/dist_test.py:212)
        if self.batch_size > 1:
            return self._split_frames(example)
        meta = example.get("metadata") or [None] * len(self.ret)
        n = self._rows if (self.point_keys and self._rows is not None and self._rows < self.capacity) else None

        def cut(v):  # per-point outputs of a padded frame: the frame's own rows
            if n is not None and torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == self.capacity:
                v = v[:n]
            return v.clone() if (clone and torch.is_tensor(v)) else v
        out = self.ret if (not clone and n is None) else [{k: cut(v) for k, v in r.items()} for r in self.ret]
        for i, r in enumerate(out):
            r["metadata"] = meta[i] if i < len(meta) else None
        return out

    def _split_frames(self, example):
        """the list predict() returns for a batch (point_seg_batchloss_head.py:255-270): labels of frame i = the rows with batch index i"""
        b = example["points"][:, 0]
        labels = self.ret[0]["pred_point_sem_labels"][:b.shape[0]]  # a padded batch: the padding rows sit behind the last frame
        meta = example.get("metadata") or [None] * self.batch_size
        out = []
        for i in range(self.batch_size):
            m = b == i
            r = dict(metadata=meta[i] if i < len(meta) else None, pred_point_sem_labels=labels[m])
            if "point_sem_labels" in example:
                r["point_sem_labels"] = example["point_sem_labels"][m]
            out.append(r)
        return out


class BucketedFrameGraph(object):
    """One FrameGraph per point-count bucket: a frame of n points runs on the graph captured for ceil(n / bucket_points) * bucket_points rows,
    padded inside the graph's input buffers (FrameGraph, `point_keys`).  Padding costs the per-point tail only (the voxelizer rejects the rows,
    the neighbour search and the decoder never visit them), so coarse buckets are cheap: 16384 points = 14 % of a 120k sweep at most ~0.1 ms.
    A bucket is captured the first time a frame falls into it (from that frame, padded); the buckets share one private memory pool; beyond
    `max_graphs` the least recently used bucket is dropped (its graph is retired, see _RETIRED).  Same results as the eager forward for every real point (tests: bit-identical labels)."""

    def __init__(self, model, bucket_points=16384, point_keys=("points", "points_cuv"), warmup=3, max_graphs=8, share_pool=True):
        self.model, self.bucket_points, self.point_keys, self.warmup, self.max_graphs = model, int(bucket_points), tuple(point_keys), warmup, int(max_graphs)
        self.graphs, self.captures = {}, 0
        # one private memory pool for all buckets: they are never replayed side by side, so a bucket's intermediates may live where another bucket's
        # were (its OUTPUTS stay allocated while its FrameGraph lives; like every clone=False result they are valid until the next call)
        self.pool = torch.cuda.graph_pool_handle() if (share_pool and torch.cuda.is_available()) else None

    def bucket(self, n):
        return max(1, -(-int(n) // self.bucket_points)) * self.bucket_points

    def _key(self, example):
        n = example["points"].shape[0]
        other = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(example.items()) if torch.is_tensor(v) and k not in self.point_keys)
        return (self.bucket(n), int(example.get("batch_size", 1)), other)

    def graph_for(self, example):
        key = self._key(example)
        fg = self.graphs.pop(key, None)
        if fg is None:
            cap, bs = key[0], key[1]
            padded = dict(example)
            for k in self.point_keys:
                if k in example:
                    padded[k] = torch.cat([example[k], pad_rows(k, example[k], cap - example[k].shape[0], bs)])
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            fg = FrameGraph(self.model, padded, warmup=self.warmup, point_keys=[k for k in self.point_keys if k in example], pool=self.pool)
            self.captures += 1
        self.graphs[key] = fg  # most recently used last
        return fg

    def __call__(self, example, clone=True):
        return self.graph_for(example)(example, clone=clone)

    @property
    def fallbacks(self):
        return sum(g.fallbacks for g in self.graphs.values())

    @property
    def recaptures(self):
        return sum(g.recaptures for g in self.graphs.values())
