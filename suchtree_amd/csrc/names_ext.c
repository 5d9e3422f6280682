/* names_ext.c -- host-side helper of SuchTree.distances_by_name (the reference's
 * name -> id loop, SuchTree/MuchTree.pyx:960-977): a list of (str, str) tuples becomes an
 * int64 (n, 2) id array with two dict lookups per pair at C speed.  Anything unexpected (an
 * item that is not a 2-tuple, a key that is not in the dict, a value that is not an int)
 * makes it return -1 with no exception set, and the Python loop in suchtree.py -- which owns
 * the reference's error messages -- runs instead.  No GPU involved; optional (the facade
 * falls back to its Python form when the module is not built). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *lookup_pairs(PyObject *self, PyObject *args)
{
    PyObject *pairs, *leaves;
    Py_buffer out;
    (void)self;
    if (!PyArg_ParseTuple(args, "O!O!w*", &PyList_Type, &pairs, &PyDict_Type, &leaves, &out)) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    long rc = 0;
    if (out.len < (Py_ssize_t)(n * 2 * (Py_ssize_t)sizeof(int64_t)) || (((uintptr_t)out.buf) & 7)) {
        PyBuffer_Release(&out);
        PyErr_SetString(PyExc_ValueError, "output buffer too small or misaligned");
        return NULL;
    }
    int64_t *dst = (int64_t *)out.buf;
    for (Py_ssize_t k = 0; k < n && rc == 0; k++) {
        /* tuples and strings are separate heap objects, visited once: the loop is all cache
         * misses unless they are requested ahead (the tuple 16 items ahead, its strings 8 ahead) */
        if (k + 16 < n) __builtin_prefetch(PyList_GET_ITEM(pairs, k + 16));
        if (k + 8 < n) {
            PyObject *ahead = PyList_GET_ITEM(pairs, k + 8);
            if (PyTuple_CheckExact(ahead) && PyTuple_GET_SIZE(ahead) == 2) {
                __builtin_prefetch(PyTuple_GET_ITEM(ahead, 0));
                __builtin_prefetch(PyTuple_GET_ITEM(ahead, 1));
            }
        }
        PyObject *item = PyList_GET_ITEM(pairs, k);
        if (!PyTuple_CheckExact(item) || PyTuple_GET_SIZE(item) != 2) { rc = -1; break; }
        for (int c = 0; c < 2; c++) {
            PyObject *name = PyTuple_GET_ITEM(item, c);
            if (!PyUnicode_CheckExact(name)) { rc = -1; break; }
            PyObject *v = PyDict_GetItemWithError(leaves, name);      /* borrowed */
            if (!v || !PyLong_CheckExact(v)) { PyErr_Clear(); rc = -1; break; }
            const long long id = PyLong_AsLongLong(v);
            if (id == -1 && PyErr_Occurred()) { PyErr_Clear(); rc = -1; break; }
            dst[2 * k + c] = (int64_t)id;
        }
    }
    PyBuffer_Release(&out);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"lookup_pairs", lookup_pairs, METH_VARARGS,
     "lookup_pairs(pairs: list[tuple[str, str]], leaves: dict[str, int], out: writable int64 buffer) -> 0, or -1 if the Python loop must take over"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_names", "name -> id lookups for distances_by_name", -1, methods,
                                       NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__names(void) { return PyModule_Create(&moduledef); }
