//! `#[model("path.tflite")]` for the MI355X build (SOURCE ONLY, never compiled here).
//!
//! Same attribute, same generated method signatures as microflow-macros/src/lib.rs:185-203.
//! Instead of emitting one `microflow::ops::*` call per operator with `const` weights, the
//! expansion embeds the .tflite bytes and forwards to libmicroflow_amd.so, which performs the
//! model reading + constant preparation the reference does at compile time and runs the ops as
//! HIP kernels.  Input/output shapes are read from the model at expansion time with the same
//! ~60-line FlatBuffers cursor as csrc/tflite.cpp (elided below as `model_io_shapes`).
extern crate proc_macro;
use proc_macro::TokenStream;
use quote::quote;
use syn::{parse_macro_input, ItemStruct, LitStr};

mod shapes; // fn model_io_shapes(bytes: &[u8]) -> Result<(Vec<usize>, Vec<usize>, bool), String>

#[proc_macro_attribute]
pub fn model(args: TokenStream, item: TokenStream) -> TokenStream {
    let path = parse_macro_input!(args as LitStr);
    let item = parse_macro_input!(item as ItemStruct);
    let ident = &item.ident;
    // The reference reads the file at expansion time relative to the crate root (the compiler's working
    // directory under cargo, microflow-macros/src/lib.rs:50).  `include_bytes!` below would resolve a
    // relative path against the INVOKING SOURCE FILE instead, so the same absolute path is used for both.
    let full = {
        let given = std::path::PathBuf::from(path.value());
        if given.is_absolute() {
            given
        } else {
            std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap_or_else(|_| ".".into())).join(given)
        }
    };
    let bytes = std::fs::read(&full).unwrap_or_else(|_| {
        panic!("couldn't find '{}', please provide a valid path", path.value()) // lib.rs:50-55
    });
    let (ishape, oshape, is_u8) = shapes::model_io_shapes(&bytes)
        .unwrap_or_else(|e| panic!("{e}")); // "invalid model, ..." / "unsupported ..." (lib.rs:56-96,148)
    // element type of predict_quantized's input, like microflow-macros/src/lib.rs:71-78
    let (qty, qcall) = if is_u8 { (quote!(u8), quote!(predict_quantized_u8)) } else { (quote!(i8), quote!(predict_quantized)) };
    let (ibuf, flatten) = match ishape.len() {
        2 => (quote!(Buffer2D), quote!(flatten_2d)),
        _ => (quote!(Buffer4D), quote!(flatten_4d)),
    };
    let (obuf, unflatten) = match oshape.len() {
        2 => (quote!(Buffer2D), quote!(unflatten_2d)),
        _ => (quote!(Buffer4D), quote!(unflatten_4d)),
    };
    let p = full.to_string_lossy().into_owned(); // absolute: the same file the shapes were read from
    quote! {
        #item
        impl #ident {
            /// ONE replica set per model type -- replica d lives on GPU d; `predict` uses replica 0,
            /// `predict_batch` all of them (no second copy of the model on GPU 0).
            fn replicas() -> std::sync::MutexGuard<'static, microflow_amd::ModelSet> {
                static S: once_cell::sync::Lazy<std::sync::Mutex<microflow_amd::ModelSet>> =
                    once_cell::sync::Lazy::new(|| std::sync::Mutex::new(
                        microflow_amd::ModelSet::new(include_bytes!(#p))));
                S.lock().unwrap()
            }
            pub fn predict(input: microflow_amd::buffer::#ibuf<f32, #(#ishape),*>)
                -> microflow_amd::buffer::#obuf<f32, #(#oshape),*> {
                let v = microflow_amd::layout::#flatten(&input);
                microflow_amd::layout::#unflatten(&Self::replicas().first().predict(&v, 1))
            }
            pub fn predict_quantized(input: microflow_amd::buffer::#ibuf<#qty, #(#ishape),*>)
                -> microflow_amd::buffer::#obuf<f32, #(#oshape),*> {
                let v = microflow_amd::layout::#flatten(&input);
                microflow_amd::layout::#unflatten(&Self::replicas().first().#qcall(&v, 1))
            }
            /// New surface: B independent inferences, sharded over every visible GPU
            /// (one launch sequence per device; no collective).
            pub fn predict_batch(inputs: &[microflow_amd::buffer::#ibuf<f32, #(#ishape),*>])
                -> Vec<microflow_amd::buffer::#obuf<f32, #(#oshape),*>> {
                if inputs.is_empty() {
                    return Vec::new();
                }
                let mut v = Vec::new();
                for i in inputs { v.extend(microflow_amd::layout::#flatten(i)); }
                let out = Self::replicas().predict(&v, inputs.len());
                let n = out.len() / inputs.len();
                out.chunks(n).map(|c| microflow_amd::layout::#unflatten(c)).collect()
            }
            /// `predict_quantized` over B independent inferences (element type i8 or u8, as the model's), sharded the same way:
            /// the batched counterpart of the reference's second generated method (microflow-macros/src/lib.rs:193-196).
            pub fn predict_quantized_batch(inputs: &[microflow_amd::buffer::#ibuf<#qty, #(#ishape),*>])
                -> Vec<microflow_amd::buffer::#obuf<f32, #(#oshape),*>> {
                if inputs.is_empty() {
                    return Vec::new();
                }
                let mut v = Vec::new();
                for i in inputs { v.extend(microflow_amd::layout::#flatten(i)); }
                let out = Self::replicas().#qcall(&v, inputs.len());
                let n = out.len() / inputs.len();
                out.chunks(n).map(|c| microflow_amd::layout::#unflatten(c)).collect()
            }
        }
    }
    .into()
}
