//! Rust shim over the C ABI of `include/microflow_amd.h`.
//!
//! NOTE: this crate is SOURCE ONLY in this repository -- the build image has no Rust
//! toolchain, so it has never been compiled here.  It shows exactly what a maintainer binds:
//! every `extern "C"` item below is declared, with the reference item it replaces, in
//! `include/microflow_amd.h`.
//!
//! Surface kept identical to MicroFlow (microflow-macros/src/lib.rs:185-203):
//! `#[model("x.tflite")] struct M;` then `M::predict(buffer)` / `M::predict_quantized(buffer)`;
//! added: `M::predict_batch(&[input]) -> Vec<output>` for many independent inferences.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub use microflow_amd_macros::model;

pub mod buffer {
    //! Same aliases as microflow::buffer (src/buffer.rs:5-16).
    pub type Buffer2D<T, const R: usize, const C: usize> = nalgebra::SMatrix<T, R, C>;
    pub type Buffer4D<T, const B: usize, const R: usize, const C: usize, const CH: usize> =
        [Buffer2D<[T; CH], R, C>; B];
}

#[repr(C)]
pub struct mf_model {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct mf_model_info {
    pub input_rank: c_int,
    pub input_shape: [c_int; 4],
    pub output_rank: c_int,
    pub output_shape: [c_int; 4],
    pub input_scale: f32,
    pub output_scale: f32,
    pub input_zero_point: c_int,
    pub output_zero_point: c_int,
    pub input_elems: usize,
    pub output_elems: usize,
    pub num_ops: c_int,
    /// 0 = i8 model, 1 = u8 model (`mf_elem_type`); the macro picks `Buffer<i8>` or `Buffer<u8>`
    /// for `predict_quantized` from it, like microflow-macros/src/lib.rs:118-128 does.
    pub element_type: c_int,
}

pub const MF_OK: c_int = 0;
pub const MF_MEM_HOST: c_int = 0;

extern "C" {
    pub fn mf_last_error() -> *const c_char;
    pub fn mf_model_create(tflite: *const u8, len: usize, model: *mut *mut mf_model) -> c_int;
    pub fn mf_model_destroy(model: *mut mf_model);
    pub fn mf_model_get_info(model: *const mf_model, info: *mut mf_model_info) -> c_int;
    pub fn mf_model_prepare(model: *mut mf_model, device: c_int, max_batch: usize) -> c_int;
    pub fn mf_model_predict(model: *mut mf_model, input: *const f32, batch: usize, output: *mut f32, mem: c_int) -> c_int;
    pub fn mf_model_predict_quantized(model: *mut mf_model, input: *const i8, batch: usize, output: *mut f32, mem: c_int) -> c_int;
    pub fn mf_model_run_quantized(model: *mut mf_model, input: *const i8, batch: usize, output: *mut i8, mem: c_int) -> c_int;
    pub fn mf_model_set_stream(model: *mut mf_model, stream: *mut c_void) -> c_int;
    pub fn mf_device_count() -> c_int;
    pub fn mf_models_predict(models: *const *mut mf_model, n_models: c_int, input: *const f32, batch: usize,
                             output: *mut f32) -> c_int;
    pub fn mf_models_predict_quantized(models: *const *mut mf_model, n_models: c_int, input: *const i8, batch: usize,
                                       output: *mut f32) -> c_int;
}

/// One prepared model on one GPU.  Not `Sync`: the C handle is single-threaded; the macro
/// wraps it in a `Mutex` inside a `once_cell::sync::Lazy`.
pub struct Model {
    raw: *mut mf_model,
    pub info: mf_model_info,
}
unsafe impl Send for Model {}

fn check(status: c_int) {
    if status != MF_OK {
        // Device errors have no compile-time counterpart in the reference: panic, like an
        // out-of-bounds index would.
        let msg = unsafe { std::ffi::CStr::from_ptr(mf_last_error()) }.to_string_lossy().into_owned();
        panic!("microflow-amd: status {status}: {msg}");
    }
}

impl Model {
    /// `bytes` = the .tflite embedded with `include_bytes!` by the macro.  Model errors (the reference's
    /// compile-time `abort_call_site!` cases) panic with the same texts; a device that cannot be prepared
    /// is reported to the caller.
    pub fn try_new(bytes: &'static [u8], device: i32) -> Result<Self, String> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { mf_model_create(bytes.as_ptr(), bytes.len(), &mut raw) });
        let mut info = mf_model_info::default();
        check(unsafe { mf_model_get_info(raw, &mut info) });
        let m = Model { raw, info }; // from here on Drop releases the handle
        let st = unsafe { mf_model_prepare(m.raw, device, 1) };
        if st != MF_OK {
            return Err(unsafe { std::ffi::CStr::from_ptr(mf_last_error()) }.to_string_lossy().into_owned());
        }
        Ok(m)
    }
    pub fn new(bytes: &'static [u8], device: i32) -> Self {
        Self::try_new(bytes, device).unwrap_or_else(|e| panic!("microflow-amd: device {device}: {e}"))
    }
    /// `input`: batch x input_elems in the ABI layout (row-major / NHWC).
    pub fn predict(&mut self, input: &[f32], batch: usize) -> Vec<f32> {
        assert_eq!(input.len(), batch * self.info.input_elems);
        let mut out = vec![0f32; batch * self.info.output_elems];
        check(unsafe { mf_model_predict(self.raw, input.as_ptr(), batch, out.as_mut_ptr(), MF_MEM_HOST) });
        out
    }
    pub fn predict_quantized(&mut self, input: &[i8], batch: usize) -> Vec<f32> {
        assert_eq!(input.len(), batch * self.info.input_elems);
        let mut out = vec![0f32; batch * self.info.output_elems];
        check(unsafe { mf_model_predict_quantized(self.raw, input.as_ptr(), batch, out.as_mut_ptr(), MF_MEM_HOST) });
        out
    }
    /// The same entry point for a UINT8 model (`info.element_type == 1`): the ABI's `int8_t*`
    /// parameter carries the raw u8 bytes.
    pub fn predict_quantized_u8(&mut self, input: &[u8], batch: usize) -> Vec<f32> {
        assert_eq!(self.info.element_type, 1);
        assert_eq!(input.len(), batch * self.info.input_elems);
        let mut out = vec![0f32; batch * self.info.output_elems];
        check(unsafe {
            mf_model_predict_quantized(self.raw, input.as_ptr() as *const i8, batch, out.as_mut_ptr(), MF_MEM_HOST)
        });
        out
    }
}
/// One replica of the model per visible GPU, driven from this one process: `predict` cuts the
/// batch into contiguous shards, one per device (mf_models_predict; no collective).
pub struct ModelSet {
    replicas: Vec<Model>,
}
impl ModelSet {
    /// One replica on every GPU that can be prepared; a device that fails (busy, out of memory) is left
    /// out instead of taking the process down.  Panics only when NO device works.
    pub fn new(bytes: &'static [u8]) -> Self {
        let n = unsafe { mf_device_count() }.max(1);
        let mut errors = Vec::new();
        let replicas: Vec<Model> = (0..n)
            .filter_map(|d| Model::try_new(bytes, d).map_err(|e| errors.push(format!("device {d}: {e}"))).ok())
            .collect();
        if replicas.is_empty() {
            panic!("microflow-amd: no usable GPU ({})", errors.join("; "));
        }
        ModelSet { replicas }
    }
    /// the replica single inferences run on
    pub fn first(&mut self) -> &mut Model {
        &mut self.replicas[0]
    }
    pub fn predict(&mut self, input: &[f32], batch: usize) -> Vec<f32> {
        let info = self.replicas[0].info;
        assert_eq!(input.len(), batch * info.input_elems);
        let mut out = vec![0f32; batch * info.output_elems];
        if batch == 0 {
            return out;
        }
        let raws: Vec<*mut mf_model> = self.replicas.iter().map(|m| m.raw).collect();
        check(unsafe { mf_models_predict(raws.as_ptr(), raws.len() as c_int, input.as_ptr(), batch, out.as_mut_ptr()) });
        out
    }
    /// `predict_quantized` (microflow-macros/src/lib.rs:193-196) over a batch, sharded like `predict`: the quantized inputs go
    /// straight to `predict_inner` on every device, only the outputs are dequantized.
    pub fn predict_quantized(&mut self, input: &[i8], batch: usize) -> Vec<f32> {
        let info = self.replicas[0].info;
        assert_eq!(info.element_type, 0);
        self.predict_quantized_raw(input.as_ptr(), input.len(), batch)
    }
    /// The same for a UINT8 model: the ABI's `int8_t*` parameter carries the raw u8 bytes.
    pub fn predict_quantized_u8(&mut self, input: &[u8], batch: usize) -> Vec<f32> {
        let info = self.replicas[0].info;
        assert_eq!(info.element_type, 1);
        self.predict_quantized_raw(input.as_ptr() as *const i8, input.len(), batch)
    }
    fn predict_quantized_raw(&mut self, input: *const i8, len: usize, batch: usize) -> Vec<f32> {
        let info = self.replicas[0].info;
        assert_eq!(len, batch * info.input_elems);
        let mut out = vec![0f32; batch * info.output_elems];
        if batch == 0 {
            return out;
        }
        let raws: Vec<*mut mf_model> = self.replicas.iter().map(|m| m.raw).collect();
        check(unsafe { mf_models_predict_quantized(raws.as_ptr(), raws.len() as c_int, input, batch, out.as_mut_ptr()) });
        out
    }
}

impl Drop for Model {
    fn drop(&mut self) {
        unsafe { mf_model_destroy(self.raw) }
    }
}

/// Layout conversion at the boundary: nalgebra `SMatrix` is column-major,
/// `Buffer4D` is `[batch][col][row][chan]` in memory; the ABI is row-major / NHWC.
pub mod layout {
    use super::buffer::{Buffer2D, Buffer4D};
    pub fn flatten_2d<T: Copy + nalgebra::Scalar, const R: usize, const C: usize>(b: &Buffer2D<T, R, C>) -> Vec<T> {
        let mut v = Vec::with_capacity(R * C);
        for i in 0..R {
            for j in 0..C {
                v.push(b[(i, j)]);
            }
        }
        v
    }
    pub fn flatten_4d<T: Copy + nalgebra::Scalar, const B: usize, const R: usize, const C: usize, const CH: usize>(
        b: &Buffer4D<T, B, R, C, CH>,
    ) -> Vec<T> {
        let mut v = Vec::with_capacity(B * R * C * CH);
        for m in b.iter() {
            for i in 0..R {
                for j in 0..C {
                    v.extend_from_slice(&m[(i, j)]);
                }
            }
        }
        v
    }
    pub fn unflatten_2d<const R: usize, const C: usize>(v: &[f32]) -> Buffer2D<f32, R, C> {
        Buffer2D::from_fn(|i, j| v[i * C + j])
    }
    pub fn unflatten_4d<const B: usize, const R: usize, const C: usize, const CH: usize>(
        v: &[f32],
    ) -> Buffer4D<f32, B, R, C, CH> {
        core::array::from_fn(|b| Buffer2D::from_fn(|i, j| core::array::from_fn(|c| v[((b * R + i) * C + j) * CH + c])))
    }
}
