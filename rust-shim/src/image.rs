//! `kmeans_color_gpu::image` -- the image container of the public API (reference `core/src/image.rs:5-64`):
//! tightly packed, row-major RGBA8, `(width, height)` in pixels.
use std::ops::Deref;

use rgb::RGBA8;

/// Anything that derefs to a pixel slice can back an [`Image`].
pub trait Container: Deref<Target = [RGBA8]> + Sized {
    fn to_pixel_vec(self) -> Vec<u8> {
        bytemuck::cast_slice::<RGBA8, u8>(&self).to_vec()
    }
}

impl Container for Vec<RGBA8> {
    fn to_pixel_vec(self) -> Vec<u8> {
        bytemuck::cast_vec(self)
    }
}

impl Container for &[RGBA8] {}

pub struct Image<C: Container> {
    pub(crate) dimensions: (u32, u32),
    pub(crate) rgba: C,
}

impl<C: Container> Image<C> {
    pub fn new(dimensions: (u32, u32), rgba: C) -> Self {
        Image { dimensions, rgba }
    }

    pub fn get_pixel(&self, x: u32, y: u32) -> &RGBA8 {
        &self.rgba[(y as usize) * (self.dimensions.0 as usize) + x as usize]
    }

    pub fn dimensions(&self) -> (u32, u32) {
        self.dimensions
    }

    pub fn into_raw_pixels(self) -> Vec<u8> {
        self.rgba.to_pixel_vec()
    }

    /// Pointer handed to libkmeans_hip (4 bytes per pixel, no row padding).
    pub(crate) fn as_bytes(&self) -> &[u8] {
        bytemuck::cast_slice::<RGBA8, u8>(&self.rgba)
    }
}

pub fn copied_pixel(dimensions: (u32, u32), rgba: &[u8]) -> Image<Vec<RGBA8>> {
    Image::new(dimensions, bytemuck::cast_slice::<u8, RGBA8>(rgba).to_vec())
}

pub fn borrowed_pixel(dimensions: (u32, u32), rgba: &[u8]) -> Image<&[RGBA8]> {
    Image::new(dimensions, bytemuck::cast_slice::<u8, RGBA8>(rgba))
}
