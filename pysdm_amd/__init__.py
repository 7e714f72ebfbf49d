"""pysdm_amd -- MI355X (gfx950) implementation of the SDM collision / coalescence / breakup hot
path: hand-written HIP kernels in libsdm_hip.so behind the C ABI of include/sdm_hip.h, and a
small host layer over it.

    abi / engine      header-driven binding; arrays + calls for one implementation of the header
    population        the super-droplet state as device columns
    recipe            collision set-ups as data (fused descriptors and pair programs)
    collisions        CollisionRunner: fused (one call per time step) and stage-by-stage routes
    displacement      DisplacementRunner (the step before collisions in 1/2/3-D)
    diagnostics       moments on the device
    sharding          cells of a multi-cell domain over the GPUs of a node (RCCL)
    cases             BASELINE.json's configurations
    backends.HIP      the same library behind PySDM's backend interface; pysdm_plugin plugs it in
"""
__all__ = ["abi", "engine", "population", "recipe", "collisions", "displacement", "diagnostics",
           "cases"]
__version__ = "0.2.0"
