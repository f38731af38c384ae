// probe (round 5): is the 48.5 TFLOP/s "sustained f64 MFMA rate" of mfma_rate.hip a clock effect?
//   * every wave stamps s_memtime (clock64: counts SHADER clocks) and s_memrealtime (wall_clock64: constant 100 MHz) around its MFMA loop:
//     effective shader clock while multiplying = Δclock / Δwall × 100 MHz — measured on the device, per wave, no driver telemetry needed;
//   * each configuration runs for >= 2 s; a host thread samples whatever the box exposes under /sys/class/drm/card*/device (hwmon freq1_input,
//     power1_average / power1_input, pp_dpm_sclk) every 50 ms beside it;
//   * DUTY < 100: the wave multiplies for DUTY % of a period and sleeps the rest (s_sleep) — what the matrix pipe does inside k_dense_epoch,
//     where a workgroup multiplies 60 % of its step.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock.bin scripts/probe/mfma_clock.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* stamps, int outer, int inner, int sleep_units) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  unsigned long long busy = 0;
  for (int o = 0; o < outer; ++o) {
    const unsigned long long t0 = clock64();
    for (int it = 0; it < inner; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    // (the accumulators must be complete before the clock is read: a dependent move)
    asm volatile("" : "+v"(acc[0]));
    busy += clock64() - t0;
    for (int s = 0; s < sleep_units; ++s) __builtin_amdgcn_s_sleep(127);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* st = stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    st[0] = c1 - c0; st[1] = w1 - w0; st[2] = busy; st[3] = 0;
  }
}

static std::atomic<bool> g_stop{false};
static std::string slurp(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r"); if (!f) return "";
  char buf[512]; size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0; return buf;
}
struct Sample { double t; long freq_hz; long power_uw; };
static void sampler(std::vector<Sample>* out, std::vector<std::string> freq_files, std::vector<std::string> power_files) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!g_stop.load()) {
    Sample s{std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), 0, 0};
    for (auto& f : freq_files) { long v = atol(slurp(f).c_str()); if (v > s.freq_hz) s.freq_hz = v; }
    for (auto& f : power_files) { long v = atol(slurp(f).c_str()); if (v > s.power_uw) s.power_uw = v; }
    out->push_back(s);
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}

template <int NACC>
void run(int blocks, int inner, int sleep_units, double seconds, const std::vector<std::string>& ff, const std::vector<std::string>& pf) {
  const int waves = blocks * 4;
  double* d; (void)hipMalloc(&d, sizeof(double) * blocks * 256);
  unsigned long long* st; (void)hipMalloc(&st, sizeof(unsigned long long) * waves * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  // calibrate: how many outer iterations make `seconds`
  k<NACC><<<blocks, 256>>>(d, st, 8, inner, sleep_units);
  (void)hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, st, 64, inner, sleep_units);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  int outer = (int)(seconds * 1e3 / ms * 64); if (outer < 64) outer = 64;
  std::vector<Sample> samples; g_stop = false;
  std::thread th(sampler, &samples, ff, pf);
  (void)hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, st, outer, inner, sleep_units);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  g_stop = true; th.join();
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(waves * 4);
  (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * waves * 4, hipMemcpyDeviceToHost);
  double mhz_sum = 0, mhz_min = 1e30, mhz_max = 0, busy_sum = 0;
  for (int w = 0; w < waves; ++w) {
    const double mhz = (double)h[w * 4] / (double)h[w * 4 + 1] * 100.0;
    mhz_sum += mhz; if (mhz < mhz_min) mhz_min = mhz; if (mhz > mhz_max) mhz_max = mhz;
    busy_sum += (double)h[w * 4 + 2] / (double)h[w * 4];
  }
  const double flops = 2.0 * 1024 * NACC * (double)inner * (double)outer * waves;
  const double tf = flops / ms / 1e9, mhz = mhz_sum / waves, duty = busy_sum / waves;
  // cycles per MFMA and SIMD while the loop runs: SIMD-cycles spent in busy periods / MFMAs issued on the SIMD
  const double waves_per_simd = (double)waves / 1024.0;
  const double cyc_per_mfma_simd = (duty * mhz * 1e6 * ms * 1e-3) / ((double)NACC * inner * (double)outer * (waves_per_simd < 1 ? 1 : waves_per_simd));
  long fmin = 0, fmax = 0, pmax = 0; double fsum = 0, psum = 0; int nf = 0, np_ = 0;
  for (auto& s : samples) {
    if (s.freq_hz > 0) { if (!nf || s.freq_hz < fmin) fmin = s.freq_hz; if (s.freq_hz > fmax) fmax = s.freq_hz; fsum += s.freq_hz; ++nf; }
    if (s.power_uw > 0) { if (s.power_uw > pmax) pmax = s.power_uw; psum += s.power_uw; ++np_; }
  }
  printf("{\"nacc\": %d, \"waves_per_simd\": %.2f, \"sleep_units\": %d, \"seconds\": %.3f, \"tflops\": %.2f, \"shader_mhz_mean\": %.1f, \"shader_mhz_min\": %.1f, "
         "\"shader_mhz_max\": %.1f, \"mfma_duty\": %.3f, \"cycles_per_mfma_per_simd_while_busy\": %.1f, \"tflops_at_2400_mhz_same_cycles\": %.2f, "
         "\"sysfs_samples\": %d, \"sysfs_sclk_mhz_mean\": %.1f, \"sysfs_sclk_mhz_min\": %.1f, \"sysfs_sclk_mhz_max\": %.1f, \"sysfs_power_w_mean\": %.1f, \"sysfs_power_w_max\": %.1f}\n",
         NACC, waves_per_simd, sleep_units, ms * 1e-3, tf, mhz, mhz_min, mhz_max, duty, cyc_per_mfma_simd, tf * 2400.0 / mhz,
         (int)samples.size(), nf ? fsum / nf / 1e6 : 0.0, fmin / 1e6, fmax / 1e6, np_ ? psum / np_ / 1e6 : 0.0, pmax / 1e6);
  fflush(stdout);
  (void)hipFree(d); (void)hipFree(st);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  std::vector<std::string> ff, pf;
  if (DIR* dd = opendir("/sys/class/drm")) {
    while (dirent* e = readdir(dd)) {
      if (strncmp(e->d_name, "card", 4) || strchr(e->d_name, '-')) continue;
      const std::string hw = std::string("/sys/class/drm/") + e->d_name + "/device/hwmon";
      if (DIR* hd = opendir(hw.c_str())) {
        while (dirent* h = readdir(hd)) {
          if (strncmp(h->d_name, "hwmon", 5)) continue;
          for (const char* n : {"freq1_input"}) { std::string p = hw + "/" + h->d_name + "/" + n; if (!slurp(p).empty()) ff.push_back(p); }
          for (const char* n : {"power1_average", "power1_input"}) { std::string p = hw + "/" + h->d_name + "/" + n; if (!slurp(p).empty()) pf.push_back(p); }
        }
        closedir(hd);
      }
    }
    closedir(dd);
  }
  fprintf(stderr, "sysfs: %zu freq files, %zu power files\n", ff.size(), pf.size());
  // every CU multiplying all the time, 1 / 2 / 4 / 8 waves per SIMD
  run<4>(256, 256, 0, seconds, ff, pf);
  run<4>(512, 256, 0, seconds, ff, pf);
  run<4>(1024, 256, 0, seconds, ff, pf);
  run<4>(2048, 256, 0, seconds, ff, pf);
  run<8>(512, 256, 0, seconds, ff, pf);
  // duty-cycled: 2 waves per SIMD, multiply ~60 % / ~40 % of the time
  run<4>(512, 256, 12, seconds, ff, pf);
  run<4>(512, 256, 30, seconds, ff, pf);
  // a quarter of the chip multiplying all the time (64 CUs): power is not the limit there
  run<4>(128, 256, 0, seconds, ff, pf);
  return 0;
}
