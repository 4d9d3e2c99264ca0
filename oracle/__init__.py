"""oracle/ — CPU restatement of the DynamiCrafter denoising path (TEST INFRASTRUCTURE, not product code).

Plain PyTorch fp32 / NumPy fp64 functional code that restates, function by function, what the reference's
lvdm modules compute on the hot path (each function cites the reference file:line it follows). It exists to
check the HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; nothing
under dynamicrafter_amd/ does, and the product path raises if the HIP extension is missing rather than falling
back to this.

Parity pin: the reference itself has no tests or golden vectors. The oracle is pinned against
outputs of the reference's own modules run in the build container (tests/golden/make_golden.py imports
/root/reference with stubs for pytorch_lightning / cv2 / torchvision and writes tests/golden/*.npz);
tests/test_oracle_golden.py replays those fixtures through this package.
"""
