"""stain2stain_amd -- the stain-to-stain flow-matching hot path on MI355X (gfx950).

Drop-in replacements for the reference's torch-only network components (same signatures and
state_dict keys), computed by hand-written HIP kernels behind a C ABI
(include/stain2stain_hip.h, stain2stain_amd/csrc/*.hip).  HIP only: there is no CPU fallback.
"""
from .components import (ClassConditionalFlowUNet, FlowMatchingDecoder, FlowUNet, SegmentationDecoder, SharedEncoder,
                         TimeEmbedding)
from .flow_matching import (ClassConditionalFlowMatchingModule, ConditionalFlowMatcher,
                            ConditionalFlowMatchingModule, MaskConditionedFlowMatchingModule,
                            MultiTaskFlowMatchingModule, ROICharbonnierFlowMatchingModule,
                            GraphedVelocity, ROIWeightedFlowMatchingModule, SolverConfig, dopri5_generate, dopri5_integrate,
                            euler_generate, euler_integrate)
from . import checkpoint
from .pix2pix import (Conv4x4Stride1, Conv4x4Stride2, ConvTranspose4x4Stride2, InstanceNormLeakyReLU,
                      PatchGANDiscriminator, Pix2PixGenerator)
from .pix2pix_engine import Pix2PixTrainer
from .trainer import CFMTrainer
from .optim import FusedAdam
from .multitask_trainer import MultiTaskTrainer

__all__ = ["SharedEncoder", "FlowMatchingDecoder", "SegmentationDecoder", "TimeEmbedding", "FlowUNet",
           "ConditionalFlowMatcher", "ConditionalFlowMatchingModule", "MultiTaskFlowMatchingModule", "euler_generate", "dopri5_generate", "euler_integrate", "dopri5_integrate", "SolverConfig", "GraphedVelocity",
           "CFMTrainer", "ClassConditionalFlowUNet", "ClassConditionalFlowMatchingModule",
           "MaskConditionedFlowMatchingModule", "ROICharbonnierFlowMatchingModule", "ROIWeightedFlowMatchingModule",
           "checkpoint", "InstanceNormLeakyReLU", "Conv4x4Stride1", "Conv4x4Stride2", "ConvTranspose4x4Stride2",
           "Pix2PixGenerator", "PatchGANDiscriminator", "Pix2PixTrainer", "FusedAdam", "MultiTaskTrainer"]
