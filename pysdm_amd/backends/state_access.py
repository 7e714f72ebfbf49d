"""Access to the bookkeeping the particle-attribute manager keeps private (permutation index,
cell_start, health flag, sorted flag, number of valid super-droplets).

The fused route owns that state for the length of a time step, so it needs to read it once and
write it back.  This package's `ParticleAttributes` offers hooks for that; PySDM's own
(PySDM/impl/particle_attributes.py:13-46) holds the same fields as name-mangled members, which
is how PySDM's `Particulator` itself reaches them (particulator.py:301-313) -- used here when the
HIP backend is plugged into an unmodified PySDM front-end."""

_PREFIX = "_ParticleAttributes__"


def view(attributes):
    hook = getattr(attributes, "_fused_view", None)
    if hook is not None:
        return hook()

    def member(name):
        return getattr(attributes, _PREFIX + name)

    return {
        "idx": member("idx"),
        "cell_start": member("cell_start"),
        "healthy": member("healthy_memory"),
        "sorted": member("sorted"),
        "valid_n_sd": member("valid_n_sd"),
        "caretaker": member("cell_caretaker"),
    }


def commit(attributes, *, valid_n_sd, sorted_flag):
    hook = getattr(attributes, "_fused_commit", None)
    if hook is not None:
        hook(valid_n_sd=valid_n_sd, sorted_flag=sorted_flag)
        return
    idx = getattr(attributes, _PREFIX + "idx")
    setattr(attributes, _PREFIX + "valid_n_sd", int(valid_n_sd))
    idx.length = idx.INT(int(valid_n_sd))
    setattr(attributes, _PREFIX + "sorted", bool(sorted_flag))


def attribute_object(attributes, name):
    getter = getattr(attributes, "get_attribute_object", None)
    if getter is not None:
        return getter(name)
    return getattr(attributes, _PREFIX + "attributes")[name]


def mark_collision_outputs_updated(attributes):
    attributes.mark_updated("multiplicity")
    for key in attributes.get_extensive_attribute_keys():
        attributes.mark_updated(key)
