"""JSSP instances of the reference's tests and notebooks (data only)."""

from queasars_amd.job_shop_scheduling import Job, JobShopSchedulingProblemInstance, Machine, Operation

NOTEBOOK_PENALTIES = dict(
    max_opt_value=100, opt_all_operations_share=0.19, encoding_penalty=319, overlap_constraint_penalty=319, precedence_constraint_penalty=275
)


def unit_test_instance():
    """test/job_shop_scheduling/problem_instance.py:15-30 (2 jobs x 2 machines, crossed routes)."""
    m1, m2 = Machine("m1"), Machine("m2")
    j1 = Job("j1", (Operation("op1", "j1", m1, 1), Operation("op2", "j1", m2, 1)))
    j2 = Job("j2", (Operation("op3", "j2", m2, 1), Operation("op4", "j2", m1, 1)))
    return JobShopSchedulingProblemInstance("instance", (m1, m2), (j1, j2))


def notebook_2x3():
    """examples/evqe_jssp_optimization.ipynb cell 2: 12 qubits at makespan_limit 6."""
    m = (Machine("m0"), Machine("m1"), Machine("m2"))
    j0 = Job("j0", (Operation("j0op0", "j0", m[2], 1), Operation("j0op1", "j0", m[0], 1), Operation("j0op2", "j0", m[1], 2)))
    j1 = Job("j1", (Operation("j1op1", "j1", m[2], 2), Operation("j1op2", "j1", m[0], 1), Operation("j1op3", "j1", m[1], 1)))
    return JobShopSchedulingProblemInstance("2_jobs_3_machines_seed_121", m, (j0, j1))


def runtime_simple_instance():
    """examples/using_the_ibm_runtime.ipynb cell 2 ("Simple Instance"): 8 qubits at makespan_limit 5 (cell 6's output)."""
    m = (Machine("m0"), Machine("m1"))
    j0 = Job("j0", (Operation("j0op0", "j0", m[0], 2), Operation("j0op1", "j0", m[1], 1)))
    j1 = Job("j1", (Operation("j1op1", "j1", m[0], 1), Operation("j1op2", "j1", m[1], 2)))
    return JobShopSchedulingProblemInstance("Simple Instance", m, (j0, j1))


def small_2x2():
    """examples/evqe_jssp_small_examples.ipynb cell 4: 4 qubits at makespan_limit 3."""
    m = (Machine("m0"), Machine("m1"))
    j0 = Job("j0", (Operation("j0op0", "j0", m[0], 1), Operation("j0op1", "j0", m[1], 1)))
    j1 = Job("j1", (Operation("j1op1", "j1", m[0], 1), Operation("j1op2", "j1", m[1], 1)))
    return JobShopSchedulingProblemInstance("2_jobs_2_machines_seed_121", m, (j0, j1))


def small_asymmetric():
    """examples/evqe_jssp_small_examples.ipynb cell 23: 5 qubits at makespan_limit 4."""
    m = (Machine("m0"), Machine("m1"), Machine("m2"))
    j0 = Job("j0", (Operation("j0op1", "j0", m[0], 1), Operation("j0op2", "j0", m[1], 2)))
    j1 = Job("j1", (Operation("j1op1", "j1", m[0], 1), Operation("j1op2", "j1", m[1], 1), Operation("j1op3", "j1", m[2], 1)))
    return JobShopSchedulingProblemInstance("2_jobs_2_machines_asymmetric", m, (j0, j1))


def three_by_three():
    """3 jobs x 3 machines, unit durations, makespan limit 5 -> 18 qubits (BASELINE.json config 4's wording)."""
    m = (Machine("m0"), Machine("m1"), Machine("m2"))
    routes = ((0, 1, 2), (1, 2, 0), (2, 0, 1))
    jobs = tuple(
        Job(f"j{j}", tuple(Operation(f"op{k}", f"j{j}", m[route[k]], 1) for k in range(3))) for j, route in enumerate(routes)
    )
    return JobShopSchedulingProblemInstance("3_jobs_3_machines", m, jobs)


def three_by_three_contended():
    """3 jobs x 3 machines, unit durations, makespan limit 5 -> 18 qubits, with two jobs starting on the same machine:
    unlike :func:`three_by_three` the all-zero state (every operation as early as possible) is NOT a valid schedule."""
    m = (Machine("m0"), Machine("m1"), Machine("m2"))
    routes = ((0, 1, 2), (0, 2, 1), (1, 0, 2))
    jobs = tuple(
        Job(f"j{j}", tuple(Operation(f"op{k}", f"j{j}", m[route[k]], 1) for k in range(3))) for j, route in enumerate(routes)
    )
    return JobShopSchedulingProblemInstance("3_jobs_3_machines_contended", m, jobs)
