"""CPU oracle for the q-diffusion quantised-UNet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under q-diffusion_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker /
reported baseline — never as the thing that is shipped or measured as "the engine".

What it is: a restatement, in plain fp32 PyTorch-CPU (the same ATen library the reference's
arithmetic lives in — the reference pins nothing tighter than `pytorch`, environment.yml:9-10) and
numpy int64, of the algorithm in /root/reference/qdiff (quant_layer.py, adaptive_rounding.py,
quant_block.py, quant_model.py), the UNet definitions it wraps (ddim/models/diffusion.py,
ldm/modules/diffusionmodules/openaimodel.py, ldm/modules/attention.py) and the sampler steps
(ddim/functions/denoising.py, ldm/models/diffusion/{ddim,plms}.py).  Every function cites the
reference file:line it follows.

Pinning: the reference ships no tests and no golden vectors, so parity is pinned
against the *live reference*: tools/make_golden.py imports /root/reference in the build container,
runs it on seeded inputs and commits the outputs under tests/golden/.  tests/test_oracle_golden.py
checks every oracle function against those vectors (bit-exact where the op order is identical).
"""
