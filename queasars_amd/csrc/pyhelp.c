/* Host-side helper of the Python layer (not part of the C ABI in include/qsv.h): packs a slice of a list of parameter
 * vectors (list[list[float]], the shape BaseCircuitEvaluator.evaluate_circuits receives them in) into doubles through
 * the CPython API.  In pure Python the fastest way, array.fromlist, costs 15 ns per value -- 40 us for the 28 vectors of
 * one push of the benchmark population, which delayed the later pushes of a step by more than a kernel's length. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

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
