// intentionally empty: stands in for the CUDA / ATen header of the same name when the reference
// kernels are compiled for the CPU oracle check (see ../cuda_on_cpu.h, force-included first).
#pragma once
