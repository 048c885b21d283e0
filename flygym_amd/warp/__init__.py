"""Import-path mirror of the reference's ``flygym.warp`` package: ``GPUSimulation`` there is the batched simulation
(``src/flygym/warp/simulation.py:28``); here that role is :class:`flygym_amd.HIPSimulation`, so

    from flygym_amd.warp import GPUSimulation

is the one-line change for code written against the reference's batched path."""


def __getattr__(name):
    if name == "GPUSimulation":
        from ..simulation import HIPSimulation

        return HIPSimulation
    raise AttributeError(name)


__all__ = ["GPUSimulation"]
