/* Host-side helper of the Python layer (not part of the C ABI in include/qsv.h): packs a slice of a list of parameter
 * vectors (list[list[float]], the shape BaseCircuitEvaluator.evaluate_circuits receives them in) into doubles through
 * the CPython API.  In pure Python the fastest way, array.fromlist, costs 15 ns per value -- 40 us for the 28 vectors of
 * one push of the benchmark population, which delayed the later pushes of a step by more than a kernel's length. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/qsv.h"

/* out[0 .. n) = the values of vectors[first], vectors[first + 1], .. vectors[first + count - 1] back to back.
 * Returns n, or -1 with a Python exception set (called through ctypes.PyDLL, which re-raises it). */
Py_ssize_t qsv_pack_vectors(PyObject* vectors, Py_ssize_t first, Py_ssize_t count, double* out, Py_ssize_t capacity) {
    PyObject* outer = PySequence_Fast(vectors, "parameter_values must be a sequence of sequences");
    if (!outer) return -1;
    Py_ssize_t n = 0;
    if (first < 0 || count < 0 || first + count > PySequence_Fast_GET_SIZE(outer)) {
        PyErr_SetString(PyExc_IndexError, "slice of parameter vectors out of range");
        Py_DECREF(outer);
        return -1;
    }
    for (Py_ssize_t i = first; i < first + count; ++i) {
        PyObject* inner = PySequence_Fast(PySequence_Fast_GET_ITEM(outer, i), "a parameter vector must be a sequence of numbers");
        if (!inner) {
            Py_DECREF(outer);
            return -1;
        }
        const Py_ssize_t m = PySequence_Fast_GET_SIZE(inner);
        if (n + m > capacity) {
            PyErr_SetString(PyExc_ValueError, "parameter vectors changed length while they were being packed");
            Py_DECREF(inner);
            Py_DECREF(outer);
            return -1;
        }
        PyObject** items = PySequence_Fast_ITEMS(inner);
        for (Py_ssize_t j = 0; j < m; ++j) {
            PyObject* v = items[j];
            double d = PyFloat_CheckExact(v) ? PyFloat_AS_DOUBLE(v) : PyFloat_AsDouble(v);
            if (d == -1.0 && PyErr_Occurred()) {
                Py_DECREF(inner);
                Py_DECREF(outer);
                return -1;
            }
            out[n++] = d;
        }
        Py_DECREF(inner);
    }
    Py_DECREF(outer);
    return n;
}

/* out[0 .. n) = the first take[i] values of vectors[i] for i = first .. first + count - 1, back to back (a vector may be
 * longer than its circuit needs; a shorter one raises ValueError).  Never writes past out[capacity): a batch whose counts
 * add up to more than the caller made room for raises ValueError.  Returns n, or -1 with a Python exception set. */
static Py_ssize_t pack_exact(PyObject* vectors, Py_ssize_t first, Py_ssize_t count, const int64_t* take, double* out,
                             Py_ssize_t capacity) {
    PyObject* outer = PySequence_Fast(vectors, "parameter_values must be a sequence of sequences");
    if (!outer) return -1;
    Py_ssize_t n = 0;
    if (first < 0 || count < 0 || first + count > PySequence_Fast_GET_SIZE(outer)) {
        PyErr_SetString(PyExc_IndexError, "slice of parameter vectors out of range");
        Py_DECREF(outer);
        return -1;
    }
    for (Py_ssize_t i = first; i < first + count; ++i) {
        PyObject* vec = PySequence_Fast_GET_ITEM(outer, i);
        if (!PyList_CheckExact(vec) && !PyTuple_CheckExact(vec) && PyObject_CheckBuffer(vec)) {
            /* a contiguous vector of doubles (a NumPy row, array('d')): copied as it stands -- what the vectorised optimiser
             * loop of evqe/solver.py hands over; as a sequence it would be unpacked into one Python float per value first */
            Py_buffer view;
            if (PyObject_GetBuffer(vec, &view, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) == 0) {
                const int is_double = view.ndim == 1 && view.itemsize == (Py_ssize_t)sizeof(double) && view.format &&
                                      (strcmp(view.format, "d") == 0 || strcmp(view.format, "<d") == 0 || strcmp(view.format, "=d") == 0);
                if (is_double) {
                    const Py_ssize_t m = view.shape[0], want = (Py_ssize_t)take[i];
                    int bad = 0;
                    if (m < want) {
                        PyErr_Format(PyExc_ValueError, "circuit %zd needs %zd parameter values, got %zd", i, want, m);
                        bad = 1;
                    } else if (want < 0 || n + want > capacity) {
                        PyErr_SetString(PyExc_ValueError, "the batch needs more parameter values than its scratch buffer holds");
                        bad = 1;
                    } else {
                        memcpy(out + n, view.buf, (size_t)want * sizeof(double));
                        n += want;
                    }
                    PyBuffer_Release(&view);
                    if (bad) {
                        Py_DECREF(outer);
                        return -1;
                    }
                    continue;
                }
                PyBuffer_Release(&view);
            } else {
                PyErr_Clear();
            }
        }
        PyObject* inner = PySequence_Fast(vec, "a parameter vector must be a sequence of numbers");
        if (!inner) {
            Py_DECREF(outer);
            return -1;
        }
        const Py_ssize_t m = PySequence_Fast_GET_SIZE(inner), want = (Py_ssize_t)take[i];
        if (m < want) {
            PyErr_Format(PyExc_ValueError, "circuit %zd needs %zd parameter values, got %zd", i, want, m);
            Py_DECREF(inner);
            Py_DECREF(outer);
            return -1;
        }
        if (want < 0 || n + want > capacity) {
            PyErr_SetString(PyExc_ValueError, "the batch needs more parameter values than its scratch buffer holds");
            Py_DECREF(inner);
            Py_DECREF(outer);
            return -1;
        }
        PyObject** items = PySequence_Fast_ITEMS(inner);
        Py_ssize_t j = 0;
        /* exact floats (what an optimiser hands over) four at a time: no call, no error check, and the four loads of the
         * objects' values do not wait for each other -- 1.3 -> 0.9 ns per value, 14 -> 10 us of a 72 us step for the 10,461
         * parameters of the benchmark population (prefetching the objects a few steps ahead: slower, 1.2 ns) */
        for (; j + 4 <= want; j += 4) {
            PyObject *v0 = items[j], *v1 = items[j + 1], *v2 = items[j + 2], *v3 = items[j + 3];
            if (!(PyFloat_CheckExact(v0) & PyFloat_CheckExact(v1) & PyFloat_CheckExact(v2) & PyFloat_CheckExact(v3))) break;
            out[n] = PyFloat_AS_DOUBLE(v0);
            out[n + 1] = PyFloat_AS_DOUBLE(v1);
            out[n + 2] = PyFloat_AS_DOUBLE(v2);
            out[n + 3] = PyFloat_AS_DOUBLE(v3);
            n += 4;
        }
        for (; j < want; ++j) {
            PyObject* v = items[j];
            double d = PyFloat_CheckExact(v) ? PyFloat_AS_DOUBLE(v) : PyFloat_AsDouble(v);
            if (d == -1.0 && PyErr_Occurred()) {
                Py_DECREF(inner);
                Py_DECREF(outer);
                return -1;
            }
            out[n++] = d;
        }
        Py_DECREF(inner);
    }
    Py_DECREF(outer);
    return n;
}

/* The whole of StatevectorDevice.expectation_values after its argument checks, in one call: lay the batch out, pack and
 * push it in as many parts as the library suggests (two halves, one per HIP stream, when there is GPU work to overlap the
 * packing with; one push for a chain of short launches), wait for the results.  The GIL is released while the call waits for the handle and for the GPU.
 * In Python the same sequence costs five ctypes calls and a dozen NumPy temporaries per population: about 90 us of a
 * 345 us step on the benchmark workload.
 * Returns the library's status (0 or QSV_E_*); -100 with a Python exception set when a parameter vector is malformed.
 * counts[i] = the number of parameters circuit i needs: that many values are taken from the front of vector i (a shorter
 * vector raises ValueError).  `values` is scratch for `capacity` >= sum(counts) doubles. */
/* (exported for the CPU tests of the capacity check) */
Py_ssize_t qsv_pack_exact(PyObject* vectors, Py_ssize_t first, Py_ssize_t count, const int64_t* take, double* out,
                          Py_ssize_t capacity) {
    return pack_exact(vectors, first, count, take, out, capacity);
}

/* QSV_HOST_TIMING=1: where the host's share of a call goes (begin / packing / push / end), printed every 2000 calls */
#include <time.h>
static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int host_timing = -1;
static double t_acc[4];
static long t_calls;

static int expectation_values(qsv_t* h, Py_ssize_t n, const int* ids, const int64_t* counts, PyObject* vectors,
                              double* values, Py_ssize_t capacity, double* out, double* device_out) {
    int rc;
    if (host_timing < 0) host_timing = getenv("QSV_HOST_TIMING") != NULL;
    double t0 = host_timing ? now_us() : 0.0, t1, t_pack = 0.0, t_push = 0.0;
    Py_BEGIN_ALLOW_THREADS
    rc = qsv_eval_begin(h, (int)n, ids, counts);
    Py_END_ALLOW_THREADS
    t1 = host_timing ? now_us() : 0.0;
    if (rc) return rc;
    if (device_out && (rc = qsv_eval_set_output(h, device_out))) {
        (void)qsv_eval_end(h, out);
        return rc;
    }
    const Py_ssize_t group = qsv_group_size(h) > 0 ? qsv_group_size(h) : 1;
    /* (QSV_PUSHES = p: p pushes per population instead of two, for measurements) */
    static int env_pushes = -1;
    if (env_pushes < 0) {
        const char* env = getenv("QSV_PUSHES");
        env_pushes = env && atoi(env) > 0 ? atoi(env) : 0;
    }
    int pushes = env_pushes > 0 ? env_pushes : qsv_eval_suggested_pushes(h);
    if (pushes < 1) pushes = 2;
    if (device_out && env_pushes == 0 && n <= 96) pushes = 1;  /* (a batch that does not wait runs on one stream) */
    /* (a push may hold more evaluations than a launch group: the library cuts it into groups itself, and split
     * evaluations -- which need no resident state -- run in much larger groups than `group`) */
    Py_ssize_t step = (n + pushes - 1) / pushes > 8 ? (n + pushes - 1) / pushes : 8;
    (void)group;
    int failed = 0, py_error = 0;
    Py_ssize_t offset = 0;
    for (Py_ssize_t first = 0; first < n && !failed; first += step) {
        const Py_ssize_t count = first + step <= n ? step : n - first;
        Py_ssize_t total = 0;
        for (Py_ssize_t i = first; i < first + count; ++i) total += (Py_ssize_t)counts[i];
        /* straight into the library's staging buffer (qsv_eval_staging): no copy on the way to the kernels */
        double* dst = NULL;
        const double ta = host_timing ? now_us() : 0.0;
        if (qsv_eval_staging(h, (int)first, (int)count, &dst) != QSV_OK || !dst) dst = values + offset;
        if (pack_exact(vectors, first, count, counts, dst, dst == values + offset ? capacity - offset : total) != total) {
            if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "parameter vectors changed length while they were being packed");
            failed = py_error = 1;
            break;
        }
        const double tb = host_timing ? now_us() : 0.0;
        rc = qsv_eval_push(h, (int)first, (int)count, total > 0 ? dst : NULL);
        if (rc) failed = 1;
        offset += total;
        if (host_timing) {
            t_pack += tb - ta;
            t_push += now_us() - tb;
        }
    }
    int rc_end;
    const double t2 = host_timing ? now_us() : 0.0;
    Py_BEGIN_ALLOW_THREADS
    rc_end = qsv_eval_end(h, out);  /* must be called even after a failed push: it releases the handle */
    Py_END_ALLOW_THREADS
    if (host_timing) {
        t_acc[0] += t1 - t0;
        t_acc[1] += t_pack;
        t_acc[2] += t_push;
        t_acc[3] += now_us() - t2;
        if (++t_calls % 2000 == 0) {
            fprintf(stderr, "host timing over %ld calls (us per call): begin %.2f  pack %.2f  push %.2f  end %.2f\n", t_calls,
                    t_acc[0] / t_calls, t_acc[1] / t_calls, t_acc[2] / t_calls, t_acc[3] / t_calls);
        }
    }
    if (py_error) return -100;
    if (failed) return rc;
    return rc_end;
}

int qsv_py_expectation_values(qsv_t* h, Py_ssize_t n, const int* ids, const int64_t* counts, PyObject* vectors,
                              double* values, Py_ssize_t capacity, double* out) {
    return expectation_values(h, n, ids, counts, vectors, values, capacity, out, NULL);
}

/* The same with the results left in device memory and no wait (qsv_eval_set_output): for the sharded population, whose
 * fitness all-gather runs on the same stream right behind. */
int qsv_py_expectation_values_device(qsv_t* h, Py_ssize_t n, const int* ids, const int64_t* counts, PyObject* vectors,
                                     double* values, Py_ssize_t capacity, void* device_out) {
    return expectation_values(h, n, ids, counts, vectors, values, capacity, NULL, (double*)device_out);
}

/* Parameter values that already live in device memory (qsv_eval_push_device): `device_values` = the values of evaluation 0,
 * the batch packed back to back by `counts` (a row-major matrix: every count the row length).  begin / push(es) / end in one
 * call, nothing packed, the GIL released throughout.  device_out as above (NULL: results to `out`, waited for). */
int qsv_py_expectation_values_devparams(qsv_t* h, Py_ssize_t n, const int* ids, const int64_t* counts, const double* device_values,
                                        void* ready_event, double* out, void* device_out) {
    int rc, rc_end;
    Py_BEGIN_ALLOW_THREADS
    rc = qsv_eval_begin(h, (int)n, ids, counts);
    if (!rc) {
        if (device_out) rc = qsv_eval_set_output(h, (double*)device_out);
        if (!rc) {
            int pushes = qsv_eval_suggested_pushes(h);
            if (pushes < 1) pushes = 2;
            if (device_out && n <= 96) pushes = 1;
            const Py_ssize_t step = (n + pushes - 1) / pushes > 8 ? (n + pushes - 1) / pushes : 8;
            Py_ssize_t offset = 0;
            for (Py_ssize_t first = 0; first < n && !rc; first += step) {
                const Py_ssize_t count = first + step <= n ? step : n - first;
                Py_ssize_t total = 0;
                for (Py_ssize_t i = first; i < first + count; ++i) total += (Py_ssize_t)counts[i];
                rc = qsv_eval_push_device(h, (int)first, (int)count, total > 0 ? device_values + offset : NULL,
                                          first == 0 ? ready_event : NULL);
                offset += total;
            }
        }
        rc_end = qsv_eval_end(h, device_out ? NULL : out);  /* (must be called even after a failed push: it releases the handle) */
        if (!rc) rc = rc_end;
    }
    Py_END_ALLOW_THREADS
    return rc;
}

/* ---- the reference's calling pattern: one circuit per call from population_size threads ---------------------------------
 * (queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/selection.py:75-82, mutation.py:63-75).  The whole call in
 * ONE C function, reached as a method of an extension module (no ctypes argument conversion): circuit id out of the
 * circuit's registration table, the parameter vector onto the stack, qsv_eval_coalesced with the GIL released, a float
 * back.  eval_one(handle, serial, circuit, vector, window_us) -> float, or None when the circuit is not registered on
 * that device yet (the caller registers it and calls again).  ValueError for bad arguments, RuntimeError otherwise. */
static PyObject* name_registered = NULL;

static PyObject* eval_one(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 5) {
        PyErr_SetString(PyExc_TypeError, "eval_one(handle, serial, circuit, vector, window_us)");
        return NULL;
    }
    qsv_t* h = (qsv_t*)PyLong_AsVoidPtr(args[0]);
    if (!h) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "null handle");
        return NULL;
    }
    PyObject* table = PyObject_GetAttr(args[2], name_registered);
    if (!table) return NULL;
    PyObject* id_obj = PyDict_Check(table) ? PyDict_GetItemWithError(table, args[1]) : NULL; /* borrowed */
    if (!id_obj) {
        Py_DECREF(table);
        if (PyErr_Occurred()) return NULL;
        Py_RETURN_NONE;
    }
    const long cid = PyLong_AsLong(id_obj);
    Py_DECREF(table);
    if (cid == -1 && PyErr_Occurred()) return NULL;
    const double window_us = PyFloat_AsDouble(args[4]);
    if (window_us == -1.0 && PyErr_Occurred()) return NULL;
    PyObject* seq = PySequence_Fast(args[3], "a parameter vector must be a sequence of numbers");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    double stack[512];
    double* values = n <= 512 ? stack : (double*)malloc((size_t)n * sizeof(double));
    if (!values) {
        Py_DECREF(seq);
        return PyErr_NoMemory();
    }
    PyObject** items = PySequence_Fast_ITEMS(seq);
    for (Py_ssize_t j = 0; j < n; ++j) {
        PyObject* v = items[j];
        const double d = PyFloat_CheckExact(v) ? PyFloat_AS_DOUBLE(v) : PyFloat_AsDouble(v);
        if (d == -1.0 && PyErr_Occurred()) {
            Py_DECREF(seq);
            if (values != stack) free(values);
            return NULL;
        }
        values[j] = d;
    }
    Py_DECREF(seq);
    double out = 0.0;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = qsv_eval_coalesced(h, (int)cid, values, (int)n, window_us, &out);
    Py_END_ALLOW_THREADS
    if (values != stack) free(values);
    if (rc) {
        const char* msg = qsv_last_error(h);
        PyErr_SetString(rc == QSV_E_ARG ? PyExc_ValueError : PyExc_RuntimeError, msg && *msg ? msg : "qsv_eval_coalesced failed");
        return NULL;
    }
    return PyFloat_FromDouble(out);
}

/* same_objects(a, b) -> bool: two lists (or tuples) of the same length holding the same objects, position by position.
 * The batch metadata of StatevectorDevice is keyed by the circuits' identities; as a tuple of 64 ids built and compared in
 * Python the check cost 2.3 us of every call. */
/* qsv_py_expectation_values_devparams as a method of the extension module (no ctypes argument conversion: 3 us of a 60 us
 * step): eval_device_matrix(handle, n, ids_address, counts_address, device_values, ready_event, out_address, device_out) -> rc,
 * every argument an integer (addresses of int32 / int64 / double arrays that outlive the call; 0 = NULL). */
static PyObject* eval_device_matrix(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 8) {
        PyErr_SetString(PyExc_TypeError, "eval_device_matrix(handle, n, ids, counts, device_values, ready_event, out, device_out)");
        return NULL;
    }
    void* p[8];
    for (int i = 0; i < 8; ++i) {
        if (i == 1) continue;
        p[i] = PyLong_AsVoidPtr(args[i]);
        if (!p[i] && PyErr_Occurred()) return NULL;
    }
    const Py_ssize_t n = PyLong_AsSsize_t(args[1]);
    if (n == -1 && PyErr_Occurred()) return NULL;
    if (!p[0] || n < 0 || (n > 0 && (!p[2] || !p[3]))) {
        PyErr_SetString(PyExc_ValueError, "eval_device_matrix: null handle or arrays");
        return NULL;
    }
    const int rc = qsv_py_expectation_values_devparams((qsv_t*)p[0], n, (const int*)p[2], (const int64_t*)p[3], (const double*)p[4],
                                                       p[5], (double*)p[6], p[7]);
    return PyLong_FromLong(rc);
}

/* qsv_py_expectation_values the same way: eval_vectors(handle, n, ids_address, counts_address, vectors, scratch_address,
 * capacity, out_address) -> rc (-100: a Python exception is set and is raised instead). */
static PyObject* eval_vectors(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 8) {
        PyErr_SetString(PyExc_TypeError, "eval_vectors(handle, n, ids, counts, vectors, scratch, capacity, out)");
        return NULL;
    }
    void* h = PyLong_AsVoidPtr(args[0]);
    const Py_ssize_t n = PyLong_AsSsize_t(args[1]);
    void* ids = PyLong_AsVoidPtr(args[2]);
    void* counts = PyLong_AsVoidPtr(args[3]);
    void* scratch = PyLong_AsVoidPtr(args[5]);
    const Py_ssize_t capacity = PyLong_AsSsize_t(args[6]);
    void* out = PyLong_AsVoidPtr(args[7]);
    if (PyErr_Occurred()) return NULL;
    if (!h || n <= 0 || !ids || !counts || !scratch || !out) {
        PyErr_SetString(PyExc_ValueError, "eval_vectors: null handle or arrays");
        return NULL;
    }
    const int rc = expectation_values((qsv_t*)h, n, (const int*)ids, (const int64_t*)counts, args[4], (double*)scratch, capacity,
                                      (double*)out, NULL);
    if (rc == -100 && PyErr_Occurred()) return NULL;
    return PyLong_FromLong(rc);
}

static PyObject* same_objects(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 2) {
        PyErr_SetString(PyExc_TypeError, "same_objects(a, b)");
        return NULL;
    }
    PyObject *a = args[0], *b = args[1];
    if (!((PyList_Check(a) || PyTuple_Check(a)) && (PyList_Check(b) || PyTuple_Check(b)))) Py_RETURN_FALSE;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(a);
    if (n != PySequence_Fast_GET_SIZE(b)) Py_RETURN_FALSE;
    PyObject **x = PySequence_Fast_ITEMS(a), **y = PySequence_Fast_ITEMS(b);
    for (Py_ssize_t i = 0; i < n; ++i)
        if (x[i] != y[i]) Py_RETURN_FALSE;
    Py_RETURN_TRUE;
}

/* has_none(seq) -> bool: some element of a list or tuple IS None.  (`None in seq` compares with ==, which an element that is
 * a NumPy array answers element by element.) */
static PyObject* has_none(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 1) {
        PyErr_SetString(PyExc_TypeError, "has_none(seq)");
        return NULL;
    }
    PyObject* a = args[0];
    if (!(PyList_Check(a) || PyTuple_Check(a))) {
        PyErr_SetString(PyExc_TypeError, "has_none: a list or tuple");
        return NULL;
    }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(a);
    PyObject** x = PySequence_Fast_ITEMS(a);
    for (Py_ssize_t i = 0; i < n; ++i)
        if (x[i] == Py_None) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

static PyMethodDef helper_methods[] = {
    {"eval_device_matrix", (PyCFunction)(void (*)(void))eval_device_matrix, METH_FASTCALL,
     "eval_device_matrix(handle, n, ids, counts, device_values, ready_event, out, device_out) -> rc: a batch whose parameter values live in device memory"},
    {"eval_vectors", (PyCFunction)(void (*)(void))eval_vectors, METH_FASTCALL,
     "eval_vectors(handle, n, ids, counts, vectors, scratch, capacity, out) -> rc: a batch of host parameter vectors"},
    {"has_none", (PyCFunction)(void (*)(void))has_none, METH_FASTCALL, "has_none(seq): some element of a list or tuple is None"},
    {"same_objects", (PyCFunction)(void (*)(void))same_objects, METH_FASTCALL,
     "same_objects(a, b): two lists or tuples hold the same objects, position by position"},
    {"eval_one", (PyCFunction)(void (*)(void))eval_one, METH_FASTCALL,
     "eval_one(handle, serial, circuit, vector, window_us): one evaluation, merged in the library with other threads'"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef helper_module = {PyModuleDef_HEAD_INIT, "_qsvpyhelp", "CPython-API helper of queasars_amd", -1, helper_methods,
                                           NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__qsvpyhelp(void) {
    name_registered = PyUnicode_InternFromString("_registered");
    if (!name_registered) return NULL;
    return PyModule_Create(&helper_module);
}
