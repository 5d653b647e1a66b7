/* CPython helper: list / tuple of N equal-length str  ->  N x L bytes (latin-1 code points).
 *
 * The reference's entry point takes Python strings (flexs/landscape.py:29-45), and at N = 1e5 turning them
 * into the byte matrix the C ABI wants costs more than the GPU work: "".join + encode + the length checks
 * is ~1.1 ms in pure Python.  This walks the sequence once and memcpy's each string's 1-byte buffer.
 *
 *   pack(seqs, L, out[, start, count]) -> 0   rows of seqs[start : start + count] written to `out` (>= count*L bytes)
 *                          1           some item has a length != L                    (caller raises)
 *                          2           some character does not fit one byte           (caller raises)
 *                          3           some item is not a str                         (caller raises)
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>

static PyObject* pack(PyObject* self, PyObject* args) {
    PyObject* seqs;
    Py_ssize_t L, start = 0, count = -1;
    Py_buffer out;
    if (!PyArg_ParseTuple(args, "Onw*|nn", &seqs, &L, &out, &start, &count)) return NULL;
    PyObject* fast = PySequence_Fast(seqs, "expected a list or tuple of str");
    if (!fast) { PyBuffer_Release(&out); return NULL; }
    const Py_ssize_t total = PySequence_Fast_GET_SIZE(fast);
    if (start < 0 || start > total) start = total;
    const Py_ssize_t n = (count < 0 || start + count > total) ? total - start : count;   /* items [start, start + n) */
    PyObject** items = PySequence_Fast_ITEMS(fast) + start;
    long status = 0;
    if (n > 0 && (L < 0 || out.len < n * L)) {
        PyErr_SetString(PyExc_ValueError, "strpack.pack: output buffer too small");
        Py_DECREF(fast); PyBuffer_Release(&out);
        return NULL;
    }
    unsigned char* dst = (unsigned char*)out.buf;
    for (Py_ssize_t i = 0; i < n; ++i, dst += L) {
        PyObject* s = items[i];
        if (!PyUnicode_Check(s) || PyUnicode_READY(s) < 0) { status = 3; break; }
        if (PyUnicode_GET_LENGTH(s) != L) { status = 1; break; }
        const int kind = PyUnicode_KIND(s);
        if (kind == PyUnicode_1BYTE_KIND) {
            memcpy(dst, PyUnicode_1BYTE_DATA(s), (size_t)L);
        } else {
            const void* data = PyUnicode_DATA(s);
            for (Py_ssize_t j = 0; j < L; ++j) {
                const Py_UCS4 c = PyUnicode_READ(kind, data, j);
                if (c > 255) { status = 2; break; }
                dst[j] = (unsigned char)c;
            }
            if (status) break;
        }
    }
    Py_DECREF(fast);
    PyBuffer_Release(&out);
    return PyLong_FromLong(status);
}

static PyMethodDef methods[] = {
    {"pack", pack, METH_VARARGS, "pack(seqs, L, out) -> status (0 ok, 1 ragged, 2 non-latin-1 character, 3 not a str)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_strpack", NULL, -1, methods};

PyMODINIT_FUNC PyInit__strpack(void) { return PyModule_Create(&moduledef); }
