//! Reads just enough of the .tflite FlatBuffer to know the model's input/output shapes and to
//! reject what the reference rejects at compile time (microflow-macros/src/lib.rs:56-96,148,161).
//! Field ids: microflow-macros/flatbuffers/tflite.fbs (same table as csrc/tflite.cpp).
fn u16_at(b: &[u8], o: usize) -> Option<usize> { Some(u16::from_le_bytes(b.get(o..o + 2)?.try_into().ok()?) as usize) }
fn u32_at(b: &[u8], o: usize) -> Option<usize> { Some(u32::from_le_bytes(b.get(o..o + 4)?.try_into().ok()?) as usize) }
fn i32_at(b: &[u8], o: usize) -> Option<i64> { Some(i32::from_le_bytes(b.get(o..o + 4)?.try_into().ok()?) as i64) }

fn field(b: &[u8], table: usize, id: usize) -> Option<usize> {
    let vt = (table as i64 - i32_at(b, table)?) as usize;
    let slot = 4 + 2 * id;
    if slot + 2 > u16_at(b, vt)? { return None; }
    match u16_at(b, vt + slot)? { 0 => None, off => Some(table + off) }
}
fn indirect(b: &[u8], o: usize) -> Option<usize> { Some(o + u32_at(b, o)?) }
fn vec_elem_table(b: &[u8], v: usize, i: usize) -> Option<usize> { indirect(b, v + 4 + 4 * i) }

/// (shape, is_u8) of tensor `idx`: TensorType INT8 = 9 or UINT8 = 3 (microflow-macros/src/lib.rs:71-78)
fn tensor_shape(b: &[u8], tensors: usize, idx: usize) -> Result<(Vec<usize>, bool), String> {
    let bad = || "invalid model, please provide a valid TensorFlow Lite model".to_string();
    let t = vec_elem_table(b, tensors, idx).ok_or_else(bad)?;
    let ty = field(b, t, 1).and_then(|o| b.get(o).copied()).unwrap_or(0);
    if ty != 9 && ty != 3 {
        return Err(format!("unsupported tensor type: {ty}. Supported types are INT8 and UINT8"));
    }
    let sh = field(b, t, 0).and_then(|o| indirect(b, o)).ok_or_else(bad)?;
    let n = u32_at(b, sh).ok_or_else(bad)?;
    let mut shape: Vec<usize> = (0..n).map(|i| i32_at(b, sh + 4 + 4 * i).unwrap_or(0) as usize).collect();
    if shape.len() == 1 { shape.insert(0, 1); }            // lib.rs:67-70
    if shape.len() != 2 && shape.len() != 4 {
        return Err(format!("unsupported tensor rank: {}. Supported ranks are 2 and 4", shape.len()));
    }
    Ok((shape, ty == 3))
}

/// (input shape, output shape, element type is u8)
pub fn model_io_shapes(b: &[u8]) -> Result<(Vec<usize>, Vec<usize>, bool), String> {
    let bad = || "invalid model, please provide a valid TensorFlow Lite model".to_string();
    let model = u32_at(b, 0).ok_or_else(bad)?;
    let sgs = field(b, model, 2).and_then(|o| indirect(b, o)).ok_or_else(bad)?;
    let sg = vec_elem_table(b, sgs, 0).ok_or_else(bad)?;     // subgraph 0 only (lib.rs:62)
    let tensors = field(b, sg, 0).and_then(|o| indirect(b, o)).ok_or_else(bad)?;
    let ins = field(b, sg, 1).and_then(|o| indirect(b, o)).ok_or_else(bad)?;
    let outs = field(b, sg, 2).and_then(|o| indirect(b, o)).ok_or_else(bad)?;
    let i0 = i32_at(b, ins + 4).ok_or_else(bad)? as usize;
    let o0 = i32_at(b, outs + 4).ok_or_else(bad)? as usize;
    let (ishape, iu8) = tensor_shape(b, tensors, i0)?;
    let (oshape, ou8) = tensor_shape(b, tensors, o0)?;
    if iu8 != ou8 {
        return Err("model input and output element types differ (mixed INT8/UINT8)".to_string());
    }
    Ok((ishape, oshape, iu8))
}
