"""Golden vectors for the JSSP datatypes, produced BY THE REFERENCE ITSELF: its problem-instance, random-instance and
serialization modules are plain Python (no Qiskit) and import in the build container from /root/reference.

    python tests/golden/make_jssp_golden.py        # writes tests/golden/jssp_reference.json

Per case: the arguments of ``random_job_shop_scheduling_instance`` (queasars/job_shop_scheduling/random_problem_instances.py:50-101),
the instance it returns in the reference's JSON wire format (serialization.py:18-76; as a parsed object and as the exact
text ``json.dumps(..., indent=2)`` gives), and a handful of random schedules of that instance with what the reference's
``JobShopSchedulingResult`` says about them (``is_valid``, ``makespan``; problem_instances.py:289-427) and their wire form.
The tests (tests/test_jssp.py) hold this repository's generator, validity rules, encoder and decoder to these vectors; the
reference is not needed to run them."""
import json
import sys
from pathlib import Path
from random import Random

sys.path.insert(0, "/root/reference")
from queasars.job_shop_scheduling.problem_instances import JobShopSchedulingResult, ScheduledOperation, UnscheduledOperation  # noqa: E402
from queasars.job_shop_scheduling.random_problem_instances import random_job_shop_scheduling_instance  # noqa: E402
from queasars.job_shop_scheduling.serialization import JSSPJSONDecoder, JSSPJSONEncoder  # noqa: E402

CASES = [
    dict(instance_name="two_by_two", n_jobs=2, n_machines=2, relative_op_amount=1.0, op_duration=1, random_seed=0),
    dict(instance_name="three_by_three", n_jobs=3, n_machines=3, relative_op_amount=1.0, op_duration={1: 0.5, 2: 0.5}, random_seed=1),
    dict(instance_name="sparse", n_jobs=4, n_machines=3, relative_op_amount={0.34: 0.3, 0.67: 0.4, 1.0: 0.3},
         op_duration={1: 0.2, 2: 0.5, 3: 0.3}, random_seed=7),
    dict(instance_name="wide", n_jobs=2, n_machines=6, relative_op_amount={0.5: 0.5, 1.0: 0.5}, op_duration={1: 0.6, 4: 0.4}, random_seed=42),
    dict(instance_name="long", n_jobs=6, n_machines=2, relative_op_amount=0.5, op_duration=3, random_seed=2024),
    dict(instance_name="five_by_four", n_jobs=5, n_machines=4, relative_op_amount={0.75: 0.5, 1.0: 0.5}, op_duration={1: 0.25, 2: 0.5, 5: 0.25},
         random_seed=99),
]


def main() -> None:
    out = []
    for args in CASES:
        instance = random_job_shop_scheduling_instance(**args)
        text = json.dumps(instance, cls=JSSPJSONEncoder, indent=2)
        assert json.loads(text, cls=JSSPJSONDecoder) == instance
        rng = Random(args["random_seed"] + 1000)
        horizon = sum(op.processing_duration for job in instance.jobs for op in job.operations)
        schedules = []
        for k in range(8):
            schedule = {}
            for job in instance.jobs:
                entries, clock = [], 0
                for op in job.operations:
                    if k >= 4 and rng.random() < 0.08:
                        entries.append(UnscheduledOperation(operation=op))
                        continue
                    # (k = 0, 1, 2: the jobs one after the other with random slack -- valid; 3: every job on its own from time 0
                    # with slack -- valid only if the machines happen not to clash; later: random starts, some left out)
                    if k < 3 and not entries and schedule:
                        clock = max(e.start_time + e.operation.processing_duration for es in schedule.values() for e in es)
                    start = clock + rng.randrange(0, 3) if k < 4 else rng.randrange(0, horizon + 1)
                    entries.append(ScheduledOperation(operation=op, start_time=start))
                    clock = start + op.processing_duration
                schedule[job] = tuple(entries)
            result = JobShopSchedulingResult(problem_instance=instance, schedule=schedule)
            schedules.append({
                "start_times": [[(e.start_time if e.is_scheduled else None) for e in schedule[job]] for job in instance.jobs],
                "is_valid": result.is_valid,
                "makespan": result.makespan,
            })
            if k in (0, 5):  # (the whole result in the wire format, instance included: two per case are enough)
                schedules[-1]["wire"] = json.loads(json.dumps(result, cls=JSSPJSONEncoder))
                schedules[-1]["repr"] = repr(result)  # (what a notebook prints)
        jsonable_args = dict(args)
        for key in ("relative_op_amount", "op_duration"):  # (JSON objects have string keys: distributions as pair lists)
            if isinstance(jsonable_args[key], dict):
                jsonable_args[key] = {"distribution": [[k, v] for k, v in jsonable_args[key].items()]}
        out.append({"arguments": jsonable_args, "wire": json.loads(text), "text_indent_2": text, "repr": repr(instance),
                    "schedules": schedules})
    path = Path(__file__).resolve().parent / "jssp_reference.json"
    path.write_text(json.dumps({"source": "the reference's own modules, run by tests/golden/make_jssp_golden.py", "cases": out},
                               separators=(",", ":")) + "\n")
    print(path, sum(len(c["schedules"]) for c in out), "schedules;",
          sum(s["is_valid"] for c in out for s in c["schedules"]), "valid")


if __name__ == "__main__":
    main()
