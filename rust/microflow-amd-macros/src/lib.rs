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
    let bytes = std::fs::read(path.value()).unwrap_or_else(|_| {
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
    let p = path.value();
    quote! {
        #item
        impl #ident {
            fn handle() -> std::sync::MutexGuard<'static, microflow_amd::Model> {
                static M: once_cell::sync::Lazy<std::sync::Mutex<microflow_amd::Model>> =
                    once_cell::sync::Lazy::new(|| std::sync::Mutex::new(
                        microflow_amd::Model::new(include_bytes!(#p), 0)));
                M.lock().unwrap()
            }
            pub fn predict(input: microflow_amd::buffer::#ibuf<f32, #(#ishape),*>)
                -> microflow_amd::buffer::#obuf<f32, #(#oshape),*> {
                let v = microflow_amd::layout::#flatten(&input);
                microflow_amd::layout::#unflatten(&Self::handle().predict(&v, 1))
            }
            pub fn predict_quantized(input: microflow_amd::buffer::#ibuf<#qty, #(#ishape),*>)
                -> microflow_amd::buffer::#obuf<f32, #(#oshape),*> {
                let v = microflow_amd::layout::#flatten(&input);
                microflow_amd::layout::#unflatten(&Self::handle().#qcall(&v, 1))
            }
            fn all_devices() -> std::sync::MutexGuard<'static, microflow_amd::ModelSet> {
                static S: once_cell::sync::Lazy<std::sync::Mutex<microflow_amd::ModelSet>> =
                    once_cell::sync::Lazy::new(|| std::sync::Mutex::new(
                        microflow_amd::ModelSet::new(include_bytes!(#p))));
                S.lock().unwrap()
            }
            /// New surface: B independent inferences, sharded over every visible GPU
            /// (one launch sequence per device; no collective).
            pub fn predict_batch(inputs: &[microflow_amd::buffer::#ibuf<f32, #(#ishape),*>])
                -> Vec<microflow_amd::buffer::#obuf<f32, #(#oshape),*>> {
                let mut v = Vec::new();
                for i in inputs { v.extend(microflow_amd::layout::#flatten(i)); }
                let out = Self::all_devices().predict(&v, inputs.len());
                let n = out.len() / inputs.len().max(1);
                out.chunks(n).map(|c| microflow_amd::layout::#unflatten(c)).collect()
            }
        }
    }
    .into()
}
