"""Host-side bitstring scoring wrapper (reference: queasars/circuit_evaluation/bitstring_evaluation.py:7-57).

The scoring callable is arbitrary host Python, so it stays on the host; the GPU only supplies the samples.
"""

from __future__ import annotations

from typing import Callable


class BitstringEvaluatorException(Exception):
    """Class for exceptions caused during the bitstring evaluation."""


class BitstringEvaluator:
    """Maps bitstrings of a fixed length to floats, validating length and alphabet first."""

    def __init__(self, input_length: int, evaluation_function: Callable[[str], float]):
        self._input_length = int(input_length)
        self._evaluation_function = evaluation_function

    @property
    def input_length(self) -> int:
        return self._input_length

    def evaluate_bitstring(self, bitstring: str) -> float:
        if len(bitstring) != self._input_length:
            raise BitstringEvaluatorException(
                f"Bitstring must be of the length {self._input_length} but was of length {len(bitstring)}!"
            )
        if set(bitstring) - {"0", "1"}:
            raise BitstringEvaluatorException("Bitstring may not contain characters other than '0' or '1'!")
        return self._evaluation_function(bitstring)
