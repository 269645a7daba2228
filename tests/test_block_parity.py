"""Teacher-forced per-block parity of the HIP engine at the FULL BASELINE shapes (SURVEY.md §8c tier T2).

The whole-UNet comparison (tests/test_engine_models.py) is dominated by the chaotic amplification of round()-tie
flips through ~60 stacked blocks, so it cannot tell a fused-epilogue defect from rounding noise.  Here every block is
evaluated on its own: the CPU oracle (oracle/unet_ref.py, pinned bit-for-bit to the real reference by
tests/test_oracle_golden.py) walks the UNet once and records, for every block, the tensors that enter it and the
tensor that leaves it; each engine block (QuantResBlock, SpatialTransformer around QuantBasicTransformerBlock,
QuantAttentionBlock, QuantResnetBlock, QuantAttnBlock, the stem / down / up / head convolutions, the time-embedding
MLP — reference qdiff/quant_block.py:83-111, 190-221, 263-271, 307-386) is then fed the ORACLE's input on the GPU and
must reproduce the ORACLE's output.

Bound per block (rng = max|oracle output|; numbers in tests/block_parity_util.py BOUNDS):
  * bulk: |diff| <= 1e-4 * rng  (fp32 rounding of the epilogue only: the contractions are exact integers), outside a
    COUNTED set of elements that a quantiser tie flip inside the block moved (a flipped activation code changes the
    3x3 x Cout neighbourhood it feeds by one quantisation step): their fraction is printed and capped;
  * mean |diff| and max |diff| per block are bounded tightly enough to catch a defect of a fused epilogue (which moves
    whole tile columns / rows by far more than a quantisation step), loosely enough for sparse one-step moves;
  * the code-flip rate itself is measured at the first quantiser of every residual block (GroupNorm -> SiLU -> int8
    codes, bit-compared with the oracle's codes of the same input), printed, and must stay <= 2e-3.
"""
import pytest
import torch

from block_parity_util import run_block_parity
from golden_util import load_fixture
from test_engine_models import _resume

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cifar_tiny", "ldm_tiny", "sd_tiny", "cifar_full", "ldm_full", "sd_full",
                                  "ldm_updown_tiny",
                                  "churches_full"])
def test_blocks_teacher_forced(cuda, name):
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)
    # the SD fixtures additionally teacher-force the three sub-layers of every transformer block (at 4096
    # tokens the whole block is judged only through the reference's fp32-vs-fp64 envelope): attention output vs the
    # exact-integer oracle at test_attention_fused's bulk bound, to_out code flips counted and bounded, sub-layer outputs vs the
    # oracle's fp32 simulation (tests/block_parity_util.py::run_sublayer_parity)
    lines, failures = run_block_parity(qnn, fx, cuda, sync=torch.cuda.synchronize, sublayers=name.startswith("sd_"))
    print("\n" + "\n".join(lines))
    assert not failures, "\n".join(failures)

