"""Mixed preset pool: a ragged batch of different systems (BASELINE configs[4]: 3wrobot + 3wrobot_NI +
2tank) driven as one job.

Environments never interact, and a wave must not mix system types (the dynamics are compile-time
policies of the kernels), so the pool sorts its envs by type into homogeneous *segments*, one
:class:`~rcognita_amd.engine.Engine` per segment.  A pool tick issues one ``rcg_control_tick`` per
segment; nothing orders the segments, so each handle runs on a non-blocking HIP stream of its own
(``rcg_use_own_stream``; measured on the configs[4] shard: 6.7e8 -> 8.0e8 env.control-steps/s) unless the caller
installs streams (``set_streams``) or passes ``own_streams=False``.  Episode statistics are merged exactly as across GPUs (:func:`rcognita_amd.parallel.merge_summaries`).
Across ranks the pool is sharded *within each type* (:func:`rcognita_amd.parallel.shard_by_type`) so every
rank sees the same mix.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native as N
from .engine import Engine, EngineConfig
from .parallel import merge_summaries, shard_by_type

# the reference presets' constants (presets/main_3wrobot.py:45-47,207-215; main_3wrobot_NI.py:45-48,207-211;
# main_2tank.py:45-48,199-211): parameters, control bounds, R1 diagonal, dt, prediction step multiplier,
# observation target
PRESETS = {
    "3wrobot": dict(sys_id=N.SYS_3WROBOT, pars=[10.0, 1.0], ctrl_bnds=[[-300.0, 300.0], [-100.0, 100.0]],
                    R1=[1.0, 10.0, 1.0, 0, 0, 0, 0], dt=0.01, mult=2.0, target=None,
                    state_init=[5.0, 5.0, -3 * np.pi / 4, 0.0, 0.0]),
    "3wrobotNI": dict(sys_id=N.SYS_3WROBOT_NI, pars=[], ctrl_bnds=[[-25.0, 25.0], [-5.0, 5.0]],
                      R1=[1.0, 10.0, 1.0, 0, 0], dt=0.01, mult=1.0, target=None,
                      state_init=[5.0, 5.0, -3 * np.pi / 4]),
    "2tank": dict(sys_id=N.SYS_2TANK, pars=[18.4, 24.4, 1.3, 1.0, 0.2], ctrl_bnds=[[0.0, 1.0]],
                  R1=[10.0, 10.0, 1.0], dt=0.1, mult=2.0, target=[0.5, 0.5], state_init=[2.0, -2.0]),
}


def preset_engine_config(name: str, batch: int, **over) -> EngineConfig:
    """EngineConfig carrying one preset's constants; keyword overrides use EngineConfig's field names."""
    p = PRESETS[name]
    kw = dict(sys_id=p["sys_id"], batch=batch, pars=p["pars"], ctrl_bnds=np.array(p["ctrl_bnds"]),
              R1=np.diag(np.array(p["R1"], dtype=float)), observation_target=p["target"], dt_sim=p["dt"],
              sampling_time=p["dt"], pred_step_size=p["dt"] * p["mult"])
    kw.update(over)
    return EngineConfig(**kw)


@dataclass
class Segment:
    name: str
    engine: Engine
    lo: int  # global env range of this segment within its type
    hi: int
    off: int = 0  # first env of the segment within THIS RANK's envs of its type (several segments per type: `parts`)


class MixedPool:
    """``counts``: {preset name: global number of envs}.  With ``rank/world`` given, this process owns the
    shard ``shard_by_type(counts, rank, world)`` of every type.

    ``parts`` > 1 cuts every type's shard into that many handles of (nearly) equal size, each on a stream of its own.
    For RQL / SQL this is worth doing even with ONE system type: a tick is {critic fit, actor kernel}, the fit is bound by
    the latency of its longest active-set walk (a few waves stay busy for 50 us) and the actor kernel by HBM, so the fit of
    one part runs under the actor kernel of another - configs[2] (131072 tank envs, RQL, streamed): 0.476 ms per tick as one
    handle, 0.430 ms as two (2.75e8 -> 3.05e8 env.control-steps/s; four parts: 0.448 ms, tools/split_probe.py)."""

    def __init__(self, counts: Dict[str, int], rank: int = 0, world: int = 1, device: int = 0, dtype: str = "f32",
                 Nactor: int = 15, mode: str = "MPC", own_streams: bool = True, parts: int = 1, **over):
        from .parallel import shard_range

        self.counts = dict(counts)
        self.segments: List[Segment] = []
        spans = shard_by_type(counts, rank, world)
        buffer_size = over.pop("buffer_size", 10) if mode != "MPC" else over.pop("buffer_size", 0)
        base_id = int(over.pop("env_id_base", 0))
        for name in sorted(counts):  # deterministic segment order
            lo, hi = spans[name]
            for p in range(max(int(parts), 1)):
                a, b = shard_range(hi - lo, p, max(int(parts), 1))
                if b <= a:
                    continue
                eng = Engine(preset_engine_config(name, b - a, device=device, dtype=dtype, Nactor=Nactor, mode=mode,
                                                  buffer_size=buffer_size, env_id_base=base_id + lo + a, **over))
                if own_streams:  # nothing orders the segments: their launches overlap on streams of their own
                    eng.use_own_stream()
                self.segments.append(Segment(name, eng, lo + a, lo + b, off=a))

    @property
    def n_envs(self) -> int:
        return sum(s.hi - s.lo for s in self.segments)

    def set_streams(self, stream_ptrs: Sequence[Optional[int]]):
        """One HIP stream per segment (e.g. ``torch.cuda.Stream().cuda_stream``): the segments are independent."""
        for s, p in zip(self.segments, stream_ptrs):
            s.engine.set_stream(p)

    def set_states(self, states: Dict[str, np.ndarray]):
        """``states[name]``: [n_local, ds] initial states of this rank's envs of that type."""
        for s in self.segments:
            s.engine.set_state(states[s.name][s.off:s.off + (s.hi - s.lo)])

    def _type_rows(self, name: str) -> int:
        return sum(s.hi - s.lo for s in self.segments if s.name == name)

    def control_tick(self, K: int, cand: Optional[Dict[str, object]] = None, producer_stream: Optional[int] = 0,
                     ordered: bool = False):
        """One env.control-step for every env of the pool: generated level grid of K candidates, or per-type
        candidate tensors ``cand[name] [n_local, K, N, du]`` (numpy, torch or :class:`DeviceArray`; ``n_local`` = this
        rank's envs of that type, whatever ``parts`` is).

        Stream ordering of device-resident candidates: the segments run on streams of their own, which nothing orders
        against the stream the caller WRITES ``cand`` on.  Both directions matter and both are handled here, on the device,
        without a host wait:
          * producer -> segments: every segment first waits (``rcg_wait_stream``) for the work queued so far on
            ``producer_stream`` - a raw HIP stream handle; the default 0 is the legacy default stream; with torch, pass
            ``torch.cuda.current_stream().cuda_stream``;
          * segments -> producer: after its tick has been issued, every segment makes ``producer_stream`` wait for it
            (``rcg_release_stream``), so a caller that REFILLS ``cand`` for the next tick on ``producer_stream``, or frees
            it into a stream-ordered allocator (torch's caching allocator reuses a block on the stream it was allocated
            on), cannot overwrite rows a segment is still reading.
        ``ordered=True`` skips both edges: for a ``cand`` that is written once before the loop, kept alive and never
        modified until :meth:`synchronize` (what bench.py's streamed regimes do)."""
        device_cand = cand is not None and any(not isinstance(c, np.ndarray) for c in cand.values())
        for s in self.segments:
            if device_cand and not ordered:
                s.engine.wait_stream(producer_stream)
            c = None if cand is None else cand[s.name]
            if c is not None:
                n, n_type = s.hi - s.lo, self._type_rows(s.name)
                rows = int(c.shape[0])
                if rows != n_type:
                    raise ValueError(f"cand[{s.name!r}] has {rows} rows, this rank holds {n_type} envs of that type")
                if n != n_type:  # several handles per type: my rows of the type's tensor
                    c = c.rows(s.off, s.off + n) if hasattr(c, "rows") else c[s.off:s.off + n]
            s.engine.control_tick(c, K=K)
            if device_cand and not ordered:
                s.engine.release_stream(producer_stream)

    def wait_stream(self, producer_stream: Optional[int] = 0):
        """Every segment's next launch waits for the work queued so far on ``producer_stream`` (see control_tick)."""
        for s in self.segments:
            s.engine.wait_stream(producer_stream)

    def release_to(self, consumer_stream: Optional[int] = 0):
        """``consumer_stream``'s next work waits for everything the segments have launched so far (the reverse edge of
        :meth:`wait_stream`; see control_tick)."""
        for s in self.segments:
            s.engine.release_stream(consumer_stream)

    def synchronize(self):
        for s in self.segments:
            s.engine.synchronize()

    def episode_reset(self):
        for s in self.segments:
            s.engine.episode_reset()

    def episode_stats(self, from_accum=False):
        """(whole-pool summary, {type: summary}) for this rank's shard (a type's parts merged)."""
        per_seg, by_type = [], {}
        for s in self.segments:
            summ, _ = s.engine.episode_stats(from_accum=from_accum)
            per_seg.append(summ)
            by_type.setdefault(s.name, []).append(summ)
        return merge_summaries(per_seg), {k: merge_summaries(v) for k, v in by_type.items()}

    def close(self):
        for s in self.segments:
            s.engine.close()
