"""The hot path as the reference composes it: ``PoseShuffleNetV2.deconv_layers``
(lib/models/networks/shufflenetv2_dcn.py:286-312) -- three
[DeformConvWithOffsetScaleBoundPositive, BatchNorm2d, ReLU, Upsample x2] groups -- in fp32, or after
``quantize_shufflenetv2_dcn`` (quantize_model.py:70-82) three
[QuantDeformConvWithOffsetScaleBoundPositive, Sequential(ReLU, QuantAct), Upsample x2] groups.

Used by bench.py, the smoke test and the multi-process tests.  Data-parallel inference shards
independent image batches over one process per GPU; the only collective is a start-up broadcast
of parameters and buffers (weights, BN statistics, QuantAct ranges) from rank 0 over RCCL
(SURVEY.md section 8e).

Round 6: split by concern into the modules of this package -- common (shared pieces), distributed, hotpath
(FusedHotPath), heads (FusedHeads), backbone (FusedBackbone), serving (calibration, FrozenHotPath, FrozenBackbone),
training (GraphedTrainStep); every name stays importable from `codenet_amd.pipeline`."""
from .common import (BN_MOMENTUM, stage_shapes, DeconvLayers, build_hot_path, make_input, set_running_stat, GATHER_PER_ITEM, GATHER_PERSISTENT, OverflowFlags, ACT_PERCENTILE, WCODES_KB, DEFER_RANGE, PHASE_SCALE, PHASE_GATHER, PHASE_POINTWISE, stage_int8_codes, act_fusable, global_range_active, uniform_act_settings, algorithmic_bytes, bn_affine)      # noqa: F401
from .distributed import (broadcast_parameters, exchange_shard_sizes, gather_detections, set_global_range, shard_range)      # noqa: F401
from .hotpath import (FusedHotPath)      # noqa: F401
from .heads import (FusedHeads)      # noqa: F401
from .backbone import (FusedBackbone)      # noqa: F401
from .serving import (cover_frozen_ranges, _widen, calibrate_serving, prepare_serving, FrozenHotPath, FrozenBackbone)      # noqa: F401
from .training import (GraphedTrainStep)      # noqa: F401
from . import backbone, common, distributed, heads, hotpath, serving, training      # noqa: F401
