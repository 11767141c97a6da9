// Field output (include/mxh.h, mxh_field_*): the reference's MxIO::save layout
// (src/MxIO.cpp:94-222) - one file per eigenvector, `mx<FieldName>_<name>_vecNN.h5`, dataset "real"
// of doubles with shape (Nx+1, Ny+1, Nz+1, ncomp), zero where the component is not in the map.
//
// HDF5 is not a build dependency: libhdf5 is looked up with dlopen at run time (MX_HDF5_LIB, the
// loader path, /opt/conda/lib, /usr/lib/x86_64-linux-gnu/hdf5/serial).  Without it the same array goes to a NumPy
// `.npy` file of the same stem, and the caller is told so.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "mxh.h"
#include "mxh_internal.hpp"

namespace {

struct Hdf5 {
  void* lib = nullptr;
  bool small_hid = false;   // HDF5 1.8: hid_t is a 32-bit int
  int64_t native_double = -1;
  typedef int (*fn_open)(void);
  typedef int (*fn_ver)(unsigned*, unsigned*, unsigned*);
  typedef int64_t (*fn_fcreate)(const char*, unsigned, int64_t, int64_t);
  typedef int64_t (*fn_fopen)(const char*, unsigned, int64_t);
  typedef int (*fn_close)(int64_t);
  typedef int64_t (*fn_screate)(int, const unsigned long long*, const unsigned long long*);
  typedef int64_t (*fn_dcreate2)(int64_t, const char*, int64_t, int64_t, int64_t, int64_t, int64_t);
  typedef int64_t (*fn_dopen2)(int64_t, const char*, int64_t);
  typedef int (*fn_dwrite)(int64_t, int64_t, int64_t, int64_t, int64_t, const void*);
  typedef int (*fn_dread)(int64_t, int64_t, int64_t, int64_t, int64_t, void*);
  typedef int64_t (*fn_dgetspace)(int64_t);
  typedef int (*fn_sdims)(int64_t, unsigned long long*, unsigned long long*);
  typedef int (*fn_sndims)(int64_t);
  fn_fcreate Fcreate = nullptr; fn_fopen Fopen = nullptr; fn_close Fclose = nullptr, Sclose = nullptr, Dclose = nullptr;
  fn_screate Screate_simple = nullptr; fn_dcreate2 Dcreate2 = nullptr; fn_dopen2 Dopen2 = nullptr;
  fn_dwrite Dwrite = nullptr; fn_dread Dread = nullptr; fn_dgetspace Dget_space = nullptr;
  fn_sdims Sget_dims = nullptr; fn_sndims Sget_ndims = nullptr;
  int64_t id(int64_t v) const { return small_hid ? (int64_t)(int32_t)(v & 0xffffffff) : v; }
};

Hdf5* hdf5() {
  static Hdf5 h;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> names;
    if (const char* e = getenv("MX_HDF5_LIB")) names.push_back(e);
    for (const char* n : {"libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
                          "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.10"})
      names.push_back(n);
    if (getenv("MX_NO_HDF5")) names.clear();
    for (const std::string& n : names) {
      void* lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!lib) continue;
      Hdf5 t;
      t.lib = lib;
      auto sym = [&](const char* s) { return dlsym(lib, s); };
      Hdf5::fn_open H5open = (Hdf5::fn_open)sym("H5open");
      Hdf5::fn_ver H5ver = (Hdf5::fn_ver)sym("H5get_libversion");
      t.Fcreate = (Hdf5::fn_fcreate)sym("H5Fcreate"); t.Fopen = (Hdf5::fn_fopen)sym("H5Fopen");
      t.Fclose = (Hdf5::fn_close)sym("H5Fclose"); t.Sclose = (Hdf5::fn_close)sym("H5Sclose"); t.Dclose = (Hdf5::fn_close)sym("H5Dclose");
      t.Screate_simple = (Hdf5::fn_screate)sym("H5Screate_simple"); t.Dcreate2 = (Hdf5::fn_dcreate2)sym("H5Dcreate2");
      t.Dopen2 = (Hdf5::fn_dopen2)sym("H5Dopen2"); t.Dwrite = (Hdf5::fn_dwrite)sym("H5Dwrite"); t.Dread = (Hdf5::fn_dread)sym("H5Dread");
      t.Dget_space = (Hdf5::fn_dgetspace)sym("H5Dget_space"); t.Sget_dims = (Hdf5::fn_sdims)sym("H5Sget_simple_extent_dims");
      t.Sget_ndims = (Hdf5::fn_sndims)sym("H5Sget_simple_extent_ndims");
      void* nd = sym("H5T_NATIVE_DOUBLE_g");
      if (!H5open || !H5ver || !t.Fcreate || !t.Fopen || !t.Fclose || !t.Sclose || !t.Dclose || !t.Screate_simple || !t.Dcreate2 ||
          !t.Dopen2 || !t.Dwrite || !t.Dread || !t.Dget_space || !t.Sget_dims || !t.Sget_ndims || !nd || H5open() < 0) {
        dlclose(lib);
        continue;
      }
      unsigned maj = 0, min = 0, rel = 0;
      H5ver(&maj, &min, &rel);
      t.small_hid = (maj == 1 && min < 10);
      t.native_double = t.small_hid ? (int64_t)*(int32_t*)nd : *(int64_t*)nd;
      h = t;
      return;
    }
  });
  return h.lib ? &h : nullptr;
}

void write_npy(const std::string& path, const int64_t dims[4], const double* data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write '" + path + "'");
  char hdr[256];
  int n = snprintf(hdr, sizeof hdr, "{'descr': '<f8', 'fortran_order': False, 'shape': (%lld, %lld, %lld, %lld), }",
                   (long long)dims[0], (long long)dims[1], (long long)dims[2], (long long)dims[3]);
  std::string h(hdr, (size_t)n);
  while ((10 + h.size() + 1) % 64 != 0) h.push_back(' ');
  h.push_back('\n');
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  const unsigned short len = (unsigned short)h.size();
  fwrite(magic, 1, 8, f); fwrite(&len, 2, 1, f); fwrite(h.data(), 1, h.size(), f);
  const size_t count = (size_t)(dims[0] * dims[1] * dims[2] * dims[3]);
  if (fwrite(data, sizeof(double), count, f) != count) { fclose(f); throw std::runtime_error("short write to '" + path + "'"); }
  fclose(f);
}

}  // namespace

extern "C" int mxh_hdf5_available(void) { return hdf5() ? 1 : 0; }

extern "C" int mxh_field_write(const char* stem, const char* dataset, const int64_t dims[4], const double* data,
                               char* written, size_t written_len) {
  try {
    if (!stem || !dataset || !dims || !data) throw std::runtime_error("mxh_field_write: NULL argument");
    std::string path;
    if (Hdf5* h = hdf5()) {
      path = std::string(stem) + ".h5";
      const unsigned long long d[4] = {(unsigned long long)dims[0], (unsigned long long)dims[1], (unsigned long long)dims[2],
                                       (unsigned long long)dims[3]};
      const int64_t file = h->id(h->Fcreate(path.c_str(), 2u /* H5F_ACC_TRUNC */, 0, 0));
      if (file < 0) throw std::runtime_error("H5Fcreate failed for '" + path + "'");
      const int64_t space = h->id(h->Screate_simple(4, d, nullptr));
      const int64_t dset = h->id(h->Dcreate2(file, dataset, h->native_double, space, 0, 0, 0));
      int rc = dset < 0 ? -1 : h->Dwrite(dset, h->native_double, 0 /* H5S_ALL */, 0, 0, data);
      if (dset >= 0) h->Dclose(dset);
      if (space >= 0) h->Sclose(space);
      h->Fclose(file);
      if (rc < 0) throw std::runtime_error("HDF5 dataset write failed for '" + path + "'");
    } else {
      path = std::string(stem) + ".npy";
      write_npy(path, dims, data);
    }
    if (written && written_len) { std::strncpy(written, path.c_str(), written_len - 1); written[written_len - 1] = '\0'; }
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

// a further dataset in the file mxh_field_write made (the reference keeps "real" and "imag" of a complex field in one
// file, src/MxIO.cpp:199-222); without HDF5 the data goes to `<stem>_<dataset>.npy`
extern "C" int mxh_field_append(const char* stem, const char* dataset, const int64_t dims[4], const double* data,
                                char* written, size_t written_len) {
  try {
    if (!stem || !dataset || !dims || !data) throw std::runtime_error("mxh_field_append: NULL argument");
    std::string path;
    if (Hdf5* h = hdf5()) {
      path = std::string(stem) + ".h5";
      const unsigned long long d[4] = {(unsigned long long)dims[0], (unsigned long long)dims[1], (unsigned long long)dims[2],
                                       (unsigned long long)dims[3]};
      const int64_t file = h->id(h->Fopen(path.c_str(), 1u /* H5F_ACC_RDWR */, 0));
      if (file < 0) throw std::runtime_error("H5Fopen (read-write) failed for '" + path + "'");
      const int64_t space = h->id(h->Screate_simple(4, d, nullptr));
      const int64_t dset = h->id(h->Dcreate2(file, dataset, h->native_double, space, 0, 0, 0));
      int rc = dset < 0 ? -1 : h->Dwrite(dset, h->native_double, 0 /* H5S_ALL */, 0, 0, data);
      if (dset >= 0) h->Dclose(dset);
      if (space >= 0) h->Sclose(space);
      h->Fclose(file);
      if (rc < 0) throw std::runtime_error("HDF5 dataset write failed for '" + path + "'");
    } else {
      path = std::string(stem) + "_" + dataset + ".npy";
      write_npy(path, dims, data);
    }
    if (written && written_len) { std::strncpy(written, path.c_str(), written_len - 1); written[written_len - 1] = '\0'; }
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}

extern "C" int mxh_field_read_h5(const char* path, const char* dataset, int64_t dims[4], double* data, int64_t capacity) {
  try {
    Hdf5* h = hdf5();
    if (!h) throw std::runtime_error("mxh_field_read_h5: no HDF5 library at run time");
    const int64_t file = h->id(h->Fopen(path, 0u /* H5F_ACC_RDONLY */, 0));
    if (file < 0) throw std::runtime_error(std::string("H5Fopen failed for '") + path + "'");
    const int64_t dset = h->id(h->Dopen2(file, dataset, 0));
    if (dset < 0) { h->Fclose(file); throw std::runtime_error(std::string("no dataset '") + dataset + "'"); }
    const int64_t space = h->id(h->Dget_space(dset));
    unsigned long long d[8] = {0};
    const int nd = h->Sget_ndims(space);
    if (nd != 4) { h->Sclose(space); h->Dclose(dset); h->Fclose(file); throw std::runtime_error("dataset is not 4-dimensional"); }
    h->Sget_dims(space, d, nullptr);
    int64_t count = 1;
    for (int i = 0; i < 4; ++i) { dims[i] = (int64_t)d[i]; count *= dims[i]; }
    int rc = 0;
    if (data) {
      if (count > capacity) rc = -2;
      else rc = h->Dread(dset, h->native_double, 0, 0, 0, data);
    }
    h->Sclose(space); h->Dclose(dset); h->Fclose(file);
    if (rc == -2) throw std::runtime_error("mxh_field_read_h5: buffer too small");
    if (rc < 0) throw std::runtime_error("H5Dread failed");
    return 0;
  } catch (const std::exception& e) {
    mxh_set_error(e.what());
    return 1;
  }
}
