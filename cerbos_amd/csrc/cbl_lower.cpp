// libcerbos_lower.so - include/cerbos_lower.h: the lowering behind a C ABI, for a host that is not Python (Go over cgo).
// The lowering itself is the package's (cerbos_amd/lower, ~5k lines of Python: CEL compiler, automata, folding): this file
// embeds CPython and calls cerbos_amd.lower.embedded.lower_pb in the calling process.  Off the hot path: once per table.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>

#include "cerbos_lower.h"

namespace {

std::once_flag g_once;
std::string g_init_error;     // why the interpreter / package could not be started (empty: started)
bool g_we_initialised = false;
thread_local std::string t_stats;

char* dup_cstr(const std::string& s) {
  char* p = static_cast<char*>(malloc(s.size() + 1));
  if (p) memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

void set_error(char** error, const std::string& s) {
  if (error) *error = dup_cstr(s);
}

// <dir of this library>/.. holds the package when the library sits in cerbos_amd/ (the in-tree build); $CERBOS_AMD_ROOT overrides
std::string package_root() {
  if (const char* env = getenv("CERBOS_AMD_ROOT")) return env;
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&cbl_abi_version), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t a = p.rfind('/');
    if (a != std::string::npos) {
      p.resize(a);                       // .../cerbos_amd
      size_t b = p.rfind('/');
      return b == std::string::npos ? std::string(".") : p.substr(0, b);
    }
  }
  return ".";
}

std::string py_error_text() {
  PyObject *type = nullptr, *value = nullptr, *tb = nullptr;
  PyErr_Fetch(&type, &value, &tb);
  PyErr_NormalizeException(&type, &value, &tb);
  std::string out = "python error";
  if (value) {
    if (PyObject* s = PyObject_Str(value)) {
      if (const char* c = PyUnicode_AsUTF8(s)) out = c;
      Py_DECREF(s);
    }
    if (type) {
      if (PyObject* n = PyObject_GetAttrString(type, "__name__")) {
        if (const char* c = PyUnicode_AsUTF8(n)) out = std::string(c) + ": " + out;
        Py_DECREF(n);
      }
    }
  }
  PyErr_Clear();
  Py_XDECREF(type); Py_XDECREF(value); Py_XDECREF(tb);
  return out;
}

void start_interpreter() {
  if (!Py_IsInitialized()) {
    Py_InitializeEx(0);                  // no signal handlers: the host keeps its own
    if (!Py_IsInitialized()) { g_init_error = "Py_InitializeEx failed"; return; }
    g_we_initialised = true;
  }
  PyGILState_STATE st = g_we_initialised ? PyGILState_LOCKED : PyGILState_Ensure();
  const std::string root = package_root();
  PyObject* sys_path = PySys_GetObject("path");   // borrowed
  PyObject* entry = PyUnicode_FromString(root.c_str());
  if (sys_path && entry && !PySequence_Contains(sys_path, entry)) PyList_Insert(sys_path, 0, entry);
  Py_XDECREF(entry);
  PyObject* mod = PyImport_ImportModule("cerbos_amd.lower.embedded");
  if (!mod) g_init_error = "cannot import cerbos_amd.lower.embedded from " + root + ": " + py_error_text();
  Py_XDECREF(mod);
  if (g_we_initialised) PyEval_SaveThread();      // leave the interpreter unlocked: callers take it with PyGILState_Ensure
  else PyGILState_Release(st);
}

}  // namespace

extern "C" {

int cbl_abi_version(void) { return CBL_ABI_VERSION; }

void cbl_free(void* p) { free(p); }

char* cbl_last_stats_json(void) { return t_stats.empty() ? nullptr : dup_cstr(t_stats); }

int cbl_lower_ruletable_pb(const uint8_t* ruletable_pb, size_t len, const char* globals_json, uint32_t flags,
                           uint8_t** image, size_t* image_len, char** error) {
  return cbl_lower_ruletable_pb_stats(ruletable_pb, len, globals_json, flags, image, image_len, nullptr, error);
}

int cbl_lower_ruletable_pb_stats(const uint8_t* ruletable_pb, size_t len, const char* globals_json, uint32_t flags,
                                 uint8_t** image, size_t* image_len, char** stats_json, char** error) {
  if (error) *error = nullptr;
  if (stats_json) *stats_json = nullptr;
  if (!image || !image_len || (!ruletable_pb && len)) { set_error(error, "null argument"); return CBL_BAD_INPUT; }
  *image = nullptr; *image_len = 0;
  std::call_once(g_once, start_interpreter);
  if (!g_init_error.empty()) { set_error(error, g_init_error); return CBL_RUNTIME; }

  PyGILState_STATE st = PyGILState_Ensure();
  int status = CBL_RUNTIME;
  PyObject *mod = nullptr, *res = nullptr;
  do {
    mod = PyImport_ImportModule("cerbos_amd.lower.embedded");
    if (!mod) { set_error(error, py_error_text()); break; }
    res = PyObject_CallMethod(mod, "lower_pb", "y#zI", reinterpret_cast<const char*>(ruletable_pb), static_cast<Py_ssize_t>(len),
                              globals_json, static_cast<unsigned int>(flags));
    if (!res) { set_error(error, py_error_text()); break; }
    long code = -1; PyObject *payload = nullptr, *stats = nullptr;
    if (!PyTuple_Check(res) || PyTuple_Size(res) != 3) { set_error(error, "unexpected result from the lowering"); break; }
    code = PyLong_AsLong(PyTuple_GetItem(res, 0));
    payload = PyTuple_GetItem(res, 1);
    stats = PyTuple_GetItem(res, 2);
    if (code == CBL_OK) {
      char* data = nullptr; Py_ssize_t n = 0;
      if (PyBytes_AsStringAndSize(payload, &data, &n) != 0) { set_error(error, py_error_text()); break; }
      uint8_t* out = static_cast<uint8_t*>(malloc(n ? static_cast<size_t>(n) : 1));
      if (!out) { set_error(error, "out of memory"); break; }
      memcpy(out, data, static_cast<size_t>(n));
      *image = out; *image_len = static_cast<size_t>(n);
      if (const char* s = PyUnicode_AsUTF8(stats)) { t_stats = s; if (stats_json) *stats_json = dup_cstr(t_stats); } else PyErr_Clear();
      status = CBL_OK;
    } else {
      const char* msg = PyUnicode_Check(payload) ? PyUnicode_AsUTF8(payload) : nullptr;
      set_error(error, msg ? msg : "lowering failed");
      status = (code == CBL_CANNOT_LOWER || code == CBL_BAD_INPUT) ? static_cast<int>(code) : CBL_RUNTIME;
    }
  } while (false);
  Py_XDECREF(res); Py_XDECREF(mod);
  PyGILState_Release(st);
  return status;
}

// ---- PlanResources (cerbos_amd/plan) behind the same interpreter
static int call_embedded(const char* fn, PyObject* args, char** error, PyObject** out_payload) {
  // -> status; *out_payload = new reference to the second element of the result tuple on CBL_OK
  int status = CBL_RUNTIME;
  PyObject *mod = nullptr, *f = nullptr, *res = nullptr;
  do {
    mod = PyImport_ImportModule("cerbos_amd.lower.embedded");
    if (!mod) { set_error(error, py_error_text()); break; }
    f = PyObject_GetAttrString(mod, fn);
    if (!f) { set_error(error, py_error_text()); break; }
    res = PyObject_CallObject(f, args);
    if (!res) { set_error(error, py_error_text()); break; }
    if (!PyTuple_Check(res) || PyTuple_Size(res) != 2) { set_error(error, "unexpected result from the planner"); break; }
    const long code = PyLong_AsLong(PyTuple_GetItem(res, 0));
    PyObject* payload = PyTuple_GetItem(res, 1);
    if (code == CBL_OK) { Py_INCREF(payload); *out_payload = payload; status = CBL_OK; }
    else {
      const char* msg = PyUnicode_Check(payload) ? PyUnicode_AsUTF8(payload) : nullptr;
      set_error(error, msg ? msg : "planner failed");
      status = code == CBL_BAD_INPUT ? CBL_BAD_INPUT : CBL_RUNTIME;
    }
  } while (false);
  Py_XDECREF(res); Py_XDECREF(f); Py_XDECREF(mod);
  return status;
}

int cbl_planner_open(const uint8_t* ruletable_pb, size_t len, uint64_t* planner, char** error) {
  if (error) *error = nullptr;
  if (!planner || (!ruletable_pb && len)) { set_error(error, "null argument"); return CBL_BAD_INPUT; }
  *planner = 0;
  std::call_once(g_once, start_interpreter);
  if (!g_init_error.empty()) { set_error(error, g_init_error); return CBL_RUNTIME; }
  PyGILState_STATE st = PyGILState_Ensure();
  PyObject* args = Py_BuildValue("(y#)", reinterpret_cast<const char*>(ruletable_pb), static_cast<Py_ssize_t>(len));
  PyObject* payload = nullptr;
  int status = args ? call_embedded("planner_open", args, error, &payload) : CBL_RUNTIME;
  if (status == CBL_OK) { *planner = (uint64_t)PyLong_AsUnsignedLongLong(payload); Py_DECREF(payload); }
  Py_XDECREF(args);
  PyGILState_Release(st);
  return status;
}

void cbl_planner_close(uint64_t planner) {
  if (!planner || !g_init_error.empty() || !Py_IsInitialized()) return;
  PyGILState_STATE st = PyGILState_Ensure();
  PyObject* args = Py_BuildValue("(K)", (unsigned long long)planner);
  PyObject* payload = nullptr;
  if (args && call_embedded("planner_close", args, nullptr, &payload) == CBL_OK) Py_XDECREF(payload);
  PyErr_Clear();
  Py_XDECREF(args);
  PyGILState_Release(st);
}

int cbl_planner_plan_pb(uint64_t planner, const uint8_t* input_pb, size_t len, const char* params_json, uint8_t** output_pb, size_t* output_len,
                        char** error) {
  if (error) *error = nullptr;
  if (!output_pb || !output_len || (!input_pb && len)) { set_error(error, "null argument"); return CBL_BAD_INPUT; }
  *output_pb = nullptr; *output_len = 0;
  if (!g_init_error.empty() || !Py_IsInitialized()) { set_error(error, g_init_error.empty() ? "no planner was opened" : g_init_error); return CBL_RUNTIME; }
  PyGILState_STATE st = PyGILState_Ensure();
  PyObject* args = Py_BuildValue("(Ky#z)", (unsigned long long)planner, reinterpret_cast<const char*>(input_pb), static_cast<Py_ssize_t>(len), params_json);
  PyObject* payload = nullptr;
  int status = args ? call_embedded("planner_plan_pb", args, error, &payload) : CBL_RUNTIME;
  if (status == CBL_OK) {
    char* data = nullptr; Py_ssize_t n = 0;
    if (PyBytes_AsStringAndSize(payload, &data, &n) != 0) { set_error(error, py_error_text()); status = CBL_RUNTIME; }
    else {
      uint8_t* out = static_cast<uint8_t*>(malloc(n ? static_cast<size_t>(n) : 1));
      if (!out) { set_error(error, "out of memory"); status = CBL_RUNTIME; }
      else { memcpy(out, data, static_cast<size_t>(n)); *output_pb = out; *output_len = static_cast<size_t>(n); }
    }
    Py_DECREF(payload);
  }
  Py_XDECREF(args);
  PyGILState_Release(st);
  return status;
}

}  // extern "C"
