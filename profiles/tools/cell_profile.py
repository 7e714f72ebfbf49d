"""phase stamps of one workgroup of the per-cell kernel (k_cell_step2, -DCELL_PROFILE build):
    scripts/build_variant.sh cellprof -DCELL_PROFILE
    SDM_HIP_LIB=build_variants/libsdm_cellprof.so PYTHONPATH=. python profiles/tools/cell_profile.py"""
from pysdm_amd.cases import make_box
from pysdm_amd.engine import HipEngine

engine = HipEngine.get()
runner = make_box(engine, "kinematic2d")
runner.run(14)  # (one call of several steps: the cell-ordered working copy from the second step on)
engine.synchronize()
