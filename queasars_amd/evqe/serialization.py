"""Population wire format: the JSON layout QUEASARS writes for EVQE populations, so that genomes saved by a real
QUEASARS install can be fed to this backend (and back) without conversion.

Key names and nesting follow the reference's encoders (data format, restated from their output):
  queasars/minimum_eigensolvers/evqe/serialization.py:33-37            individual
  queasars/minimum_eigensolvers/evqe/serialization.py:40-66            population (species bookkeeping as lists of pairs)
  queasars/minimum_eigensolvers/evqe/quantum_circuit/serialization.py:29-59   layer and the four gate kinds

    individual  {"evqe_individual_n_qubits", "evqe_individual_layers", "evqe_individual_parameter_values"}
    layer       {"evqe_circuit_layer_n_qubits", "evqe_circuit_layer_gates"}
    gate        {"evqe_gate_type": "identity" | "rotation" | "control" | "controlled_rotation", "evqe_qubit_index",
                 "evqe_controlled_qubit_index" (control), "evqe_control_qubit_index" (controlled_rotation)}
    population  {"evqe_population_individuals", "evqe_population_species_representatives" (list | null),
                 "evqe_population_species_members" ([[individual, [indices]], ..] | null),
                 "evqe_population_species_membership" ([[index, individual], ..] | null)}
"""

from __future__ import annotations

import json
from typing import Any, Optional

from queasars_amd.evqe.genome import (
    ControlGate,
    ControlledRotationGate,
    EVQECircuitLayer,
    EVQEGate,
    EVQEGateType,
    EVQEIndividual,
    EVQEPopulation,
    IdentityGate,
    RotationGate,
)

_GATE_NAMES = {
    EVQEGateType.IDENTITY: "identity",
    EVQEGateType.ROTATION: "rotation",
    EVQEGateType.CONTROL: "control",
    EVQEGateType.CONTROLLED_ROTATION: "controlled_rotation",
}


def gate_to_dict(gate: EVQEGate) -> dict[str, Any]:
    out: dict[str, Any] = {"evqe_gate_type": _GATE_NAMES[gate.kind], "evqe_qubit_index": gate.qubit_index}
    if gate.kind is EVQEGateType.CONTROL:
        out["evqe_controlled_qubit_index"] = gate.partner_index
    elif gate.kind is EVQEGateType.CONTROLLED_ROTATION:
        out["evqe_control_qubit_index"] = gate.partner_index
    return out


def gate_from_dict(data: dict[str, Any]) -> EVQEGate:
    kind, qubit = data["evqe_gate_type"], int(data["evqe_qubit_index"])
    if kind == "identity":
        return IdentityGate(qubit)
    if kind == "rotation":
        return RotationGate(qubit)
    if kind == "control":
        return ControlGate(qubit, int(data["evqe_controlled_qubit_index"]))
    if kind == "controlled_rotation":
        return ControlledRotationGate(qubit, int(data["evqe_control_qubit_index"]))
    raise ValueError(f"unknown evqe_gate_type {kind!r}")


def layer_to_dict(layer: EVQECircuitLayer) -> dict[str, Any]:
    return {"evqe_circuit_layer_n_qubits": layer.n_qubits, "evqe_circuit_layer_gates": [gate_to_dict(g) for g in layer.gates]}


def layer_from_dict(data: dict[str, Any]) -> EVQECircuitLayer:
    return EVQECircuitLayer(
        n_qubits=int(data["evqe_circuit_layer_n_qubits"]),
        gates=tuple(gate_from_dict(g) for g in data["evqe_circuit_layer_gates"]),
    )


def individual_to_dict(individual: EVQEIndividual) -> dict[str, Any]:
    return {
        "evqe_individual_n_qubits": individual.n_qubits,
        "evqe_individual_layers": [layer_to_dict(layer) for layer in individual.layers],
        "evqe_individual_parameter_values": [float(v) for v in individual.parameter_values],
    }


def individual_from_dict(data: dict[str, Any]) -> EVQEIndividual:
    return EVQEIndividual(
        n_qubits=int(data["evqe_individual_n_qubits"]),
        layers=tuple(layer_from_dict(layer) for layer in data["evqe_individual_layers"]),
        parameter_values=tuple(float(v) for v in data["evqe_individual_parameter_values"]),
    )


def population_to_dict(population: EVQEPopulation) -> dict[str, Any]:
    reps = population.species_representatives
    members = population.species_members
    membership = population.species_membership
    return {
        "evqe_population_individuals": [individual_to_dict(i) for i in population.individuals],
        "evqe_population_species_representatives": None if reps is None else [individual_to_dict(i) for i in reps],
        "evqe_population_species_members": None
        if members is None
        else [[individual_to_dict(rep), [int(m) for m in idx]] for rep, idx in members.items()],
        "evqe_population_species_membership": None
        if membership is None
        else [[int(idx), individual_to_dict(rep)] for idx, rep in membership.items()],
    }


def population_from_dict(data: dict[str, Any]) -> EVQEPopulation:
    def optional(key: str) -> Optional[Any]:
        return data.get(key)

    reps, members, membership = (
        optional("evqe_population_species_representatives"),
        optional("evqe_population_species_members"),
        optional("evqe_population_species_membership"),
    )
    return EVQEPopulation(
        individuals=tuple(individual_from_dict(i) for i in data["evqe_population_individuals"]),
        species_representatives=None if reps is None else [individual_from_dict(i) for i in reps],
        species_members=None if members is None else {individual_from_dict(rep): [int(m) for m in idx] for rep, idx in members},
        species_membership=None if membership is None else {int(idx): individual_from_dict(rep) for idx, rep in membership},
    )


def dumps(population: EVQEPopulation, **json_kwargs: Any) -> str:
    return json.dumps(population_to_dict(population), **json_kwargs)


def loads(text: str) -> EVQEPopulation:
    return population_from_dict(json.loads(text))
