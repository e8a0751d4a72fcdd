// Source-compatibility stand-in for the reference's jxl::ThreadPool
// (/root/reference/encoder/base/data_parallel.h).  The reference constructs a
// pool in EncodeFile and passes it to EncodeFrame, but never dispatches work to
// it; this encoder runs its host stage on its own std::threads
// and its pixel pipeline on the GPU, so the type only carries a thread count.
#ifndef JXLT_HOST_ENCODER_BASE_DATA_PARALLEL_H_
#define JXLT_HOST_ENCODER_BASE_DATA_PARALLEL_H_

namespace jxl {

class ThreadPool {
 public:
  explicit ThreadPool(int num_threads = 0) : num_threads_(num_threads) {}
  int NumThreads() const { return num_threads_; }  // <= 0: all cores

 private:
  int num_threads_;
};

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_BASE_DATA_PARALLEL_H_
