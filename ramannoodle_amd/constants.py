"""Constants used on the evaluation path (``ramannoodle/constants.py:248-249``)."""

RAMAN_TENSOR_CENTRAL_DIFFERENCE = 0.001
BOLTZMANN_CONSTANT = 8.617333262e-5  # eV/K
