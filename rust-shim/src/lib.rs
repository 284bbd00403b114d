//! `kmeans_color_gpu` on AMD MI355X: the public surface of the reference crate
//! (`core/src/lib.rs:24-253`: `ImageProcessor::{new, palette, find, reduce}`, `Image`, `Algorithm`,
//! `ReduceMode`, `ColorSpace`, `RGBA8`) implemented by `libkmeans_hip.so` through `ffi`.
//!
//! The methods stay `async` so that callers that `block_on` them (cli/src/main.rs:27-40, core/examples) compile
//! unchanged; the futures are ready immediately -- the library call is synchronous and thread safe
//! (one `ImageProcessor` may be shared by many threads, core/examples/parallel.rs:36-50).
use std::ffi::CStr;
use std::fmt::Display;
use std::os::raw::c_int;
use std::str::FromStr;

use anyhow::{anyhow, Result};
pub use rgb::RGBA8;

use crate::image::{Container, Image};

mod ffi;
pub mod image;

pub struct ImageProcessor {
    raw: *mut ffi::kmg_processor,
    /// non-null: the processor spans several devices (`with_devices`); the calls go through `kmg_group_*`
    group: *mut ffi::kmg_group,
}

// every entry point of libkmeans_hip is re-entrant on one processor (per-call stream and workspace)
unsafe impl Send for ImageProcessor {}
unsafe impl Sync for ImageProcessor {}

fn check(rc: c_int) -> Result<()> {
    if rc == ffi::KMG_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::kmg_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("kmeans_hip error {rc}: {msg}"))
}

impl ImageProcessor {
    /// lib.rs:38-65.  Fails when no HIP device is usable (there is no CPU path), as the reference fails
    /// without a wgpu adapter.
    pub async fn new() -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::kmg_processor_create(&mut raw) })?;
        log::debug!("{}", unsafe { CStr::from_ptr(ffi::kmg_version()) }.to_string_lossy());
        Ok(ImageProcessor { raw, group: std::ptr::null_mut() })
    }

    /// The same constructor over several GPUs of the node (HIP ordinals; empty = every visible device).  No counterpart in
    /// the reference, which picks one adapter (lib.rs:38-65): `palette`, `find` and `reduce` then tile the image in row bands
    /// over the devices and return the same bytes; a sharded Lloyd loop all-reduces its k x 4 integer sums with RCCL.
    pub async fn with_devices(devices: &[i32]) -> Result<Self> {
        let mut opt: ffi::kmg_group_options = unsafe { std::mem::zeroed() };
        unsafe { ffi::kmg_default_group_options(&mut opt) };
        if devices.len() > ffi::KMG_MAX_DEVICES {
            return Err(anyhow!("at most {} devices", ffi::KMG_MAX_DEVICES));
        }
        opt.n_devices = devices.len() as u32;
        opt.devices[..devices.len()].copy_from_slice(devices);
        let mut group = std::ptr::null_mut();
        check(unsafe { ffi::kmg_group_create(&opt, &mut group) })?;
        Ok(ImageProcessor { raw: std::ptr::null_mut(), group })
    }

    /// lib.rs:67-77: `color_count` dominant colours, sorted by Lab lightness (k-means: exactly
    /// `color_count`; octree: at most).
    pub async fn palette<C: Container>(
        &self,
        color_count: u32,
        image: &Image<C>,
        algo: Algorithm,
    ) -> Result<Vec<RGBA8>> {
        let (width, height) = image.dimensions();
        let mut out = vec![RGBA8::default(); color_count.max(1) as usize];
        let mut count = 0u32;
        let (px, dst) = (image.as_bytes().as_ptr(), out.as_mut_ptr() as *mut u8);
        check(unsafe {
            if self.group.is_null() {
                ffi::kmg_palette(self.raw, px, width, height, color_count, algo.as_c(), dst, &mut count)
            } else {
                ffi::kmg_group_palette(self.group, px, width, height, color_count, algo.as_c(), dst, &mut count)
            }
        })?;
        out.truncate(count as usize);
        Ok(out)
    }

    /// lib.rs:79-114: every pixel replaced by (or dithered / melded towards) the closest of `colors`.
    pub async fn find<C: Container>(
        &self,
        image: &Image<C>,
        colors: &[RGBA8],
        reduce_mode: &ReduceMode,
    ) -> Result<Image<Vec<RGBA8>>> {
        let (width, height) = image.dimensions();
        let mut out = vec![RGBA8::default(); width as usize * height as usize];
        let (px, pal, dst) = (image.as_bytes().as_ptr(), colors.as_ptr() as *const u8, out.as_mut_ptr() as *mut u8);
        check(unsafe {
            if self.group.is_null() {
                ffi::kmg_find(self.raw, px, width, height, pal, colors.len() as u32, reduce_mode.as_c(), dst)
            } else {
                ffi::kmg_group_find(self.group, px, width, height, pal, colors.len() as u32, reduce_mode.as_c(), dst)
            }
        })?;
        Ok(Image::new((width, height), out))
    }

    /// lib.rs:116-164: palette extraction (`algo`) followed by `find` with that palette.
    pub async fn reduce<C: Container>(
        &self,
        color_count: u32,
        image: &Image<C>,
        algo: &Algorithm,
        reduce_mode: &ReduceMode,
    ) -> Result<Image<Vec<RGBA8>>> {
        let (width, height) = image.dimensions();
        let mut out = vec![RGBA8::default(); width as usize * height as usize];
        let (px, dst) = (image.as_bytes().as_ptr(), out.as_mut_ptr() as *mut u8);
        check(unsafe {
            if self.group.is_null() {
                ffi::kmg_reduce(self.raw, px, width, height, color_count, algo.as_c(), reduce_mode.as_c(), dst)
            } else {
                ffi::kmg_group_reduce(self.group, px, width, height, color_count, algo.as_c(), reduce_mode.as_c(), dst)
            }
        })?;
        Ok(Image::new((width, height), out))
    }
}

impl Drop for ImageProcessor {
    fn drop(&mut self) {
        unsafe {
            if !self.group.is_null() {
                ffi::kmg_group_destroy(self.group)
            }
            ffi::kmg_processor_destroy(self.raw)
        }
    }
}

/// lib.rs:167-213.  Only `Lab` is reachable from the reference's public API (lib.rs:87,94,130,266); kept
/// because `cli/src/args.rs:131-137` converts into it.
#[derive(Clone, Copy)]
pub enum ColorSpace {
    Lab,
    Rgb,
}

impl ColorSpace {
    pub fn from(s: &str) -> Option<ColorSpace> {
        s.parse().ok()
    }

    pub fn name(&self) -> &'static str {
        match self {
            ColorSpace::Lab => "lab",
            ColorSpace::Rgb => "rgb",
        }
    }

    pub fn convergence(&self) -> f32 {
        match self {
            ColorSpace::Lab => 1.0,
            ColorSpace::Rgb => 0.01,
        }
    }
}

impl FromStr for ColorSpace {
    type Err = anyhow::Error;

    fn from_str(s: &str) -> Result<Self, Self::Err> {
        match s {
            "lab" => Ok(ColorSpace::Lab),
            "rgb" => Ok(ColorSpace::Rgb),
            other => Err(anyhow!("Unsupported color space {other}")),
        }
    }
}

impl Display for ColorSpace {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(self.name())
    }
}

/// lib.rs:215-232
#[derive(Clone, Copy)]
pub enum Algorithm {
    Kmeans,
    Octree,
}

impl Algorithm {
    fn as_c(&self) -> c_int {
        match self {
            Algorithm::Kmeans => ffi::KMG_ALGO_KMEANS,
            Algorithm::Octree => ffi::KMG_ALGO_OCTREE,
        }
    }
}

impl Display for Algorithm {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(match self {
            Algorithm::Kmeans => "kmeans",
            Algorithm::Octree => "octree",
        })
    }
}

/// lib.rs:234-253
#[derive(Clone, Copy)]
pub enum ReduceMode {
    Replace,
    Dither,
    Meld,
}

impl ReduceMode {
    fn as_c(&self) -> c_int {
        match self {
            ReduceMode::Replace => ffi::KMG_MODE_REPLACE,
            ReduceMode::Dither => ffi::KMG_MODE_DITHER,
            ReduceMode::Meld => ffi::KMG_MODE_MELD,
        }
    }
}

impl Display for ReduceMode {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(match self {
            ReduceMode::Replace => "replace",
            ReduceMode::Dither => "dither",
            ReduceMode::Meld => "meld",
        })
    }
}
