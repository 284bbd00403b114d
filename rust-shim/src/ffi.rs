//! `extern "C"` view of `include/kmeans_hip.h` -- the host-buffer API, i.e. exactly what
//! `ImageProcessor::{new, palette, find, reduce}` (reference `core/src/lib.rs:38-164`) need.
//! Every item below is checked against the header by `tests/test_rust_shim.py`.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int};

/// Opaque `kmg_processor` (include/kmeans_hip.h).
#[repr(C)]
pub struct kmg_processor {
    _private: [u8; 0],
}

/// `kmg_options` (include/kmeans_hip.h); `kmg_default_options` fills in the reference's constants
/// (structures.rs:23, modules.rs:765-766, lib.rs:189-194).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct kmg_options {
    pub struct_size: u32,
    pub device: i32,
    pub shrink_max_dim: u32,
    pub max_iterations: u32,
    pub check_period: u32,
    pub convergence: f32,
    pub strategy: i32,
}

/// `kmg_options.strategy` (KMG_STRATEGY_*): 0 = the library's cost models decide per call; results are identical either way.
pub const KMG_STRATEGY_AUTO: i32 = 0;
pub const KMG_STRATEGY_SCAN: i32 = 1;
pub const KMG_STRATEGY_TABLE: i32 = 2;
pub const KMG_STRATEGY_MASK_WORDS: i32 = 4;

/// Opaque `kmg_group`: a processor + RCCL rank per device of a list (include/kmeans_hip.h, "a group of devices").
#[repr(C)]
pub struct kmg_group {
    _private: [u8; 0],
}

pub const KMG_MAX_DEVICES: usize = 16;

/// `kmg_group_options` (include/kmeans_hip.h); `kmg_default_group_options` fills it in.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct kmg_group_options {
    pub struct_size: u32,
    pub n_devices: u32,
    pub devices: [i32; KMG_MAX_DEVICES],
    pub flags: u32,
    pub processor: kmg_options,
}

pub const KMG_OK: c_int = 0;
pub const KMG_ALGO_KMEANS: c_int = 0;
pub const KMG_ALGO_OCTREE: c_int = 1;
pub const KMG_MODE_REPLACE: c_int = 0;
pub const KMG_MODE_DITHER: c_int = 1;
pub const KMG_MODE_MELD: c_int = 2;

extern "C" {
    pub fn kmg_last_error() -> *const c_char;
    pub fn kmg_version() -> *const c_char;
    pub fn kmg_default_options(opt: *mut kmg_options);
    // ImageProcessor::new -- lib.rs:38-65
    pub fn kmg_processor_create(out: *mut *mut kmg_processor) -> c_int;
    pub fn kmg_processor_create_ex(opt: *const kmg_options, out: *mut *mut kmg_processor) -> c_int;
    pub fn kmg_processor_destroy(p: *mut kmg_processor);
    pub fn kmg_processor_set_strategy(p: *mut kmg_processor, strategy: c_int) -> c_int;
    // ImageProcessor::palette -- lib.rs:67-77
    pub fn kmg_palette(
        p: *mut kmg_processor,
        rgba: *const u8,
        width: u32,
        height: u32,
        color_count: u32,
        algo: c_int,
        out_rgba: *mut u8,
        out_count: *mut u32,
    ) -> c_int;
    // ImageProcessor::find -- lib.rs:79-114
    pub fn kmg_find(
        p: *mut kmg_processor,
        rgba: *const u8,
        width: u32,
        height: u32,
        palette_rgba: *const u8,
        n_colors: u32,
        mode: c_int,
        out_rgba: *mut u8,
    ) -> c_int;
    // ImageProcessor::reduce -- lib.rs:116-164
    pub fn kmg_reduce(
        p: *mut kmg_processor,
        rgba: *const u8,
        width: u32,
        height: u32,
        color_count: u32,
        algo: c_int,
        mode: c_int,
        out_rgba: *mut u8,
    ) -> c_int;
    // ImageProcessor::new over a device list (the reference is single-device: lib.rs:38-65) and the same three calls, the image
    // tiled in row bands over the devices, the k x 4 sums of a sharded Lloyd loop all-reduced by RCCL inside the library
    pub fn kmg_default_group_options(opt: *mut kmg_group_options);
    pub fn kmg_group_create(opt: *const kmg_group_options, out: *mut *mut kmg_group) -> c_int;
    pub fn kmg_group_destroy(g: *mut kmg_group);
    pub fn kmg_group_palette(
        g: *mut kmg_group,
        rgba: *const u8,
        width: u32,
        height: u32,
        color_count: u32,
        algo: c_int,
        out_rgba: *mut u8,
        out_count: *mut u32,
    ) -> c_int;
    pub fn kmg_group_find(
        g: *mut kmg_group,
        rgba: *const u8,
        width: u32,
        height: u32,
        palette_rgba: *const u8,
        n_colors: u32,
        mode: c_int,
        out_rgba: *mut u8,
    ) -> c_int;
    pub fn kmg_group_reduce(
        g: *mut kmg_group,
        rgba: *const u8,
        width: u32,
        height: u32,
        color_count: u32,
        algo: c_int,
        mode: c_int,
        out_rgba: *mut u8,
    ) -> c_int;
}
