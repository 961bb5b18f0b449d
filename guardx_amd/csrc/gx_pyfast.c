/* gx_pyfast.c -- CPython shim for the two per-step entry points of include/guardx.h.
 *
 * An unmodified learner drives Engine.step() / reset_done() once per control step from Python (safe_rl_libX/trpo/
 * trpo.py:479-547).  At env_num = 2000 that loop is host bound, and a ctypes call with seven arguments costs ~2.5 us of
 * argument conversion -- more than the kernel launch it makes.  This module calls the same C ABI functions through their
 * addresses (handed over once by guardx_amd/_native.py, which loaded libguardx_hip.so with ctypes) with plain integer
 * arguments: ~0.3 us per call.  No arithmetic lives here; without it the Engine uses ctypes, same results.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*step_slab_fn)(void*, const float*, float*, int32_t, int32_t, int32_t*, void*);
typedef int (*commit_fn)(void*);

static step_slab_fn g_step_slab = NULL;
static commit_fn g_commit = NULL;

static PyObject* bind(PyObject* self, PyObject* args)
{
    unsigned long long a = 0, b = 0;
    if (!PyArg_ParseTuple(args, "KK", &a, &b)) return NULL;
    g_step_slab = (step_slab_fn)(uintptr_t)a;
    g_commit = (commit_fn)(uintptr_t)b;
    Py_RETURN_NONE;
}

/* step_slab(handle, action_ptr, slab_ptr, slot, flags, stream) -> status | speculated << 8 */
static PyObject* step_slab(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    if (nargs != 6 || !g_step_slab) {
        PyErr_SetString(PyExc_TypeError, "step_slab(handle, action_ptr, slab_ptr, slot, flags, stream) after bind()");
        return NULL;
    }
    void* h = PyLong_AsVoidPtr(args[0]);
    const float* act = (const float*)PyLong_AsVoidPtr(args[1]);
    float* slab = (float*)PyLong_AsVoidPtr(args[2]);
    const long slot = PyLong_AsLong(args[3]);
    const long flags = PyLong_AsLong(args[4]);
    void* stream = PyLong_AsVoidPtr(args[5]);
    if (PyErr_Occurred()) return NULL;
    int32_t spec = 0;
    const int st = g_step_slab(h, act, slab, (int32_t)slot, (int32_t)flags, &spec, stream);
    return PyLong_FromLong((long)(st & 0xff) | ((long)(spec ? 1 : 0) << 8));
}

/* reset_done_commit(handle) -> status */
static PyObject* reset_done_commit(PyObject* self, PyObject* arg)
{
    if (!g_commit) { PyErr_SetString(PyExc_TypeError, "reset_done_commit() before bind()"); return NULL; }
    void* h = PyLong_AsVoidPtr(arg);
    if (PyErr_Occurred()) return NULL;
    return PyLong_FromLong((long)g_commit(h));
}

static PyMethodDef methods[] = {
    {"bind", bind, METH_VARARGS, "bind(addr of gx_step_slab, addr of gx_reset_done_commit)"},
    {"step_slab", (PyCFunction)(void (*)(void))step_slab, METH_FASTCALL, "gx_step_slab with integer arguments"},
    {"reset_done_commit", reset_done_commit, METH_O, "gx_reset_done_commit"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_gxfast", "fast calls into libguardx_hip.so", -1, methods};

PyMODINIT_FUNC PyInit__gxfast(void) { return PyModule_Create(&moddef); }
