// Drop-in for the reference's lib/nms/gpu_nms.hpp (the one native ABI the reference exports; lib/nms/gpu_nms.hpp:1-2,
// defined by nms_kernel.cu:90-143, bound by gpu_nms.pyx:10-11).  libadvmix_hip.so defines this symbol with C++ linkage
// (csrc/nms.hip): box NMS on the MI355X - one wave per 64 x 64 IoU tile - behind the reference's own argument list.
// keep_out: room for boxes_num ints; boxes_host: [boxes_num, boxes_dim = 5] fp32 rows (x1, y1, x2, y2, score) sorted by score,
// best first; *num_out receives the number of kept indices.  Synchronous; errors are printed, not returned.
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
          int boxes_dim, float nms_overlap_thresh, int device_id);
