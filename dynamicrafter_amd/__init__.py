"""dynamicrafter_amd — MI355X (gfx950) native device path for the DynamiCrafter denoising loop.

Host side mirrors the reference's ``lvdm`` classes (``dynamicrafter_amd.lvdm.*`` keeps the dotted layout of the
reference so ``configs/*.yaml`` ``target:`` strings resolve, see ``dynamicrafter_amd.registry``); every device
op goes through the C ABI in ``include/dcrafter_hip.h`` (``dynamicrafter_amd/csrc/libdcrafter_hip.so``).
"""
__version__ = "0.1.0"
