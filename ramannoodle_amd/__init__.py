"""MI355X-native evaluation of ramannoodle's PotGNN polarizability model.

Public surface (mirrors the reference for this path only):

* ``ramannoodle_amd.pmodel.PotGNN``            -- ``ramannoodle.pmodel.torch.PotGNN``
* ``ramannoodle_amd.dynamics.Phonons/Trajectory`` -- ``ramannoodle.dynamics``
* ``ramannoodle_amd.spectrum.*``               -- ``ramannoodle.spectrum``
* ``ramannoodle_amd.parallel``                 -- frame sharding across GPUs (new)
"""
from ramannoodle_amd import abstract, constants, dynamics, exceptions, spectrum, structure  # noqa: F401

__version__ = "0.1.0"
