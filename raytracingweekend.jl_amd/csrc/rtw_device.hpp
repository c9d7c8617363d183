// rtw_device.hpp -- device-side building blocks of the hot path, gfx950 (MI355X) only.
//
// Each function restates one reference function (paths relative to /root/reference) under the
// numerics contract of DESIGN.md section 4: IEEE binary32/64, one rounding per written
// operation (this file is compiled with -ffp-contract=off), explicit FMA only in the
// ray-sphere discriminant.  Float32 mode is the reference's mixed precision:
// geometry / RNG / scatter in T, sky colour, throughput and accumulation in double.
// The parts: rtw_path.hpp (everything of the path tracer but the scan), rtw_scan.hpp (device scene, all-VALU scan), rtw_scan_mfma.hpp
// (matrix-pipe scan + the group cull's table vote), rtw_scan_cull.hpp (group-cull layout, its all-VALU scan, LDS staging).
#pragma once
#include "rtw_path.hpp"
#include "rtw_scan.hpp"
#include "rtw_scan_mfma.hpp"
#include "rtw_scan_cull.hpp"
