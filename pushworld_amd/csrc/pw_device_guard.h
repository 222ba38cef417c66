// PwDeviceGuard (declared in pw_host.h): include after <hip/hip_runtime.h>.
#ifndef PW_DEVICE_GUARD_H_
#define PW_DEVICE_GUARD_H_

struct PwDeviceGuard {
  explicit PwDeviceGuard(int device) {
    err_ = hipGetDevice(&prev_);
    if (err_ != hipSuccess) {
      prev_ = -1;
      err_ = hipSuccess;  // no current device yet: nothing to restore
    }
    if (device >= 0 && device != prev_) {
      err_ = hipSetDevice(device);
      changed_ = err_ == hipSuccess;
    }
  }
  ~PwDeviceGuard() {
    if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  PwDeviceGuard(const PwDeviceGuard&) = delete;
  PwDeviceGuard& operator=(const PwDeviceGuard&) = delete;
  hipError_t status() const { return err_; }

 private:
  int prev_ = -1;
  bool changed_ = false;
  hipError_t err_ = hipSuccess;
};

#endif  // PW_DEVICE_GUARD_H_
