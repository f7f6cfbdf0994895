"""go-muse_amd: MI355X-native engine for go-muse's XCorr / Batch.Run hot path.

The directory name carries a hyphen (it mirrors the reference's name), so load
it with importlib.import_module("go-muse_amd").  Contents:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI implementation
  build.py   hipcc driver -> lib/libmuse_hip.so (in-tree)
  binding.py ctypes view of include/muse_hip.h
  muse.py    host-side mirror of the reference's Series/Group/Results/Batch API
  dist.py    one-process-per-GPU sharding + gather of per-shard top-N records
"""
from . import build  # noqa: F401
from . import binding  # noqa: F401
from .binding import MuseError  # noqa: F401
from . import dist  # noqa: F401
from .muse import (Batch, DefaultLabel, DeviceBatch, DeviceGroup, Engine, Group, Labels, Muse, New,  # noqa: F401
                   NewBatch, NewGroup, NewLabels, NewResults, NewSeries, Results, Score, Series,
                   SignFilter_ANY, SignFilter_NEG, SignFilter_POS, get_engine, merge_records, next_pow2,
                   RunMany, run_many, score_many, scores_many, xcorr_groups, device_count,
                   merge_group_records, merge_group_winners)
